/* Compiled as plain C99 by tests/test_abi.py: include/w2a.h must be usable without C++ or HIP headers, and the
 * host-only entry points must work without a GPU. */
#include <stdio.h>
#include <string.h>

#include "w2a.h"

int main(void) {
  w2a_tables t;
  w2a_policy p;
  w2a_state_view v;
  w2a_env *h = NULL;
  memset(&t, 0, sizeof t);
  memset(&p, 0, sizeof p);
  memset(&v, 0, sizeof v);
  if (w2a_abi_version() != W2A_ABI_VERSION) return 1;
  if (w2a_state_bytes(1000) < 256 + 40 * 1000 || w2a_state_bytes(1000) % 256) return 2;
  t.T = 153; t.S_w = 746; t.Y = 11; t.S = 746; t.n_samples = 100;
  if (w2a_group_workspace_bytes(1000, 10, 10) < 4 * 4 * 1000 || w2a_group_workspace_bytes(0, 10, 10) != 0) return 3;
  if (w2a_create(&t, 8, 0, NULL, 0, NULL, &h) != W2A_ERR_ARG || h != NULL) return 4;
  if (!strstr(w2a_last_error(), "NULL")) return 5;
  if (w2a_step(NULL, NULL, W2A_ACT_I32, NULL, NULL, NULL, NULL, 0, NULL) != W2A_ERR_ARG) return 6;
  if (w2a_rollout(NULL, &p, 1, NULL, NULL, NULL, NULL, NULL, 0, NULL, NULL, NULL) != W2A_ERR_ARG) return 7;
  if (w2a_policy_actions(NULL, &p, NULL, NULL, NULL, NULL, NULL, 0, NULL) != W2A_ERR_ARG) return 10;
  if (w2a_rollout_posterior_mean(NULL, &p, 1, NULL, NULL, NULL, NULL, NULL, 0, NULL, NULL, NULL) != W2A_ERR_ARG) return 13;
  if (w2a_set_semantics(NULL, W2A_FIX_ALL) != W2A_ERR_ARG) return 8;
  if (w2a_sort_workspace_bytes(0) != 0) return 9;
  if (w2a_rollout_order_workspace_bytes(0, 10) != 0 || w2a_rollout_order_workspace_bytes(1000, 10) < 4 * 1000) return 11;
  if (w2a_rollout_order(NULL, NULL, 0, NULL) != W2A_ERR_ARG) return 12;
  if (w2a_rollout_order_attach(NULL, NULL, 0) != W2A_ERR_ARG) return 14;
  if (w2a_invalidate(NULL, NULL) != W2A_ERR_ARG) return 15;
  printf("w2a C ABI v%d ok, sizeof(w2a_tables)=%zu\n", w2a_abi_version(), sizeof(w2a_tables));
  return 0;
}
