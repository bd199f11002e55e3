/* exported probes of oracle/pyset.h for tests/test_oracle_pyset.py (test infrastructure) */
#include "pyset.h"

#define NSETS 8
static pyset g_sets[NSETS];
static int g_init[NSETS];

static pyset *S(int h) {
  if (!g_init[h]) {
    pyset_init(&g_sets[h]);
    g_init[h] = 1;
  }
  return &g_sets[h];
}

void pst_clear(int h) {
  if (g_init[h]) pyset_free(&g_sets[h]);
  g_init[h] = 0;
}
void pst_add(int h, int key) { pyset_add(S(h), key); }
int pst_remove(int h, int key) { return pyset_remove(S(h), key); }
int pst_pop(int h) { return pyset_pop(S(h)); }
int pst_len(int h) { return (int)S(h)->used; }
int pst_list(int h, int32_t *out) { return (int)pyset_list(S(h), out); }
void pst_copy(int dst, int src) {
  pst_clear(dst);
  pyset_copy(&g_sets[dst], S(src));
  g_init[dst] = 1;
}
