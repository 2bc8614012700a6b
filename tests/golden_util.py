import json
import os

import numpy as np

_P = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ode_golden.json")


def fh(xs):
    return np.array([float.fromhex(x) for x in xs], dtype=np.float64)


def load_cases():
    return json.load(open(_P))["cases"]


def case_ids():
    return [c["name"] for c in load_cases()]
