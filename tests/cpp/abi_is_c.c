/* The boundary header must be consumable from plain C (that is what a Nim {.importc.} binding sees). */
#include "nnhip_ode.h"
#include <stdio.h>
int main(void) {
  nnhip_ode_options o;
  if (nnhip_ode_default_options(&o) != NNHIP_OK) return 1;
  if (nnhip_ode_new_options(&o, 1e-4, 1e-4, 1e-4, 1e-5, 1e-4, 4.0, 0.1, 0.0) != NNHIP_EVALUE) return 2; /* dtMax < dtMin */
  if (nnhip_ode_integrator_id("TSIT54") != NNHIP_TSIT54) return 3;
  if (nnhip_ode_integrator_id("rk5") != NNHIP_EINTEGRATOR) return 4;
  printf("abi %d ok: %s\n", nnhip_abi_version(), nnhip_last_error());
  return 0;
}
