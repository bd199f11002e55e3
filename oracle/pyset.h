/* oracle/pyset.h - model of CPython 3.10's `set` for small non-negative ints (hash(i) == i).
 *
 * TEST INFRASTRUCTURE (see oracle/README.md).
 *
 * Why it exists: the reference's trajectories depend on set iteration / pop order. Executor
 * pools are Python sets of executor ids (reference components/executor_tracker.py:37-42,77,90);
 * `_fulfill_commitments_from_source` pops from `set(generator over pool.copy())`
 * (spark_sched_sim.py:714-741), `_move_idle_executors` iterates `list(set)` (:762) and
 * `_compute_jobtime` sums floats in `set(list + list)` order (:855-864). CPython is the
 * reference's runtime (CI pins Python 3.10, .github/workflows/python-app.yml:22-25), not part of
 * the reference tree, so this restates the published algorithm of Objects/setobject.c
 * (open addressing, LINEAR_PROBES = 9, PERTURB_SHIFT = 5, resize at fill*5 >= mask*3 to the
 * first power of two > used*4). It is differential-tested against the live interpreter in
 * tests/test_oracle_pyset.py.
 */
#ifndef SSS_ORACLE_PYSET_H
#define SSS_ORACLE_PYSET_H

#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define PYSET_EMPTY (-1)
#define PYSET_DUMMY (-2)
#define PYSET_MINSIZE 8
#define PYSET_LINEAR_PROBES 9
#define PYSET_PERTURB_SHIFT 5

typedef struct {
  int32_t *table;
  int64_t mask; /* table size - 1 */
  int64_t fill; /* active + dummy */
  int64_t used; /* active */
  int64_t finger;
} pyset;

static inline void pyset_init(pyset *s) {
  s->table = (int32_t *)malloc(sizeof(int32_t) * PYSET_MINSIZE);
  for (int i = 0; i < PYSET_MINSIZE; i++) s->table[i] = PYSET_EMPTY;
  s->mask = PYSET_MINSIZE - 1;
  s->fill = s->used = s->finger = 0;
}

static inline void pyset_free(pyset *s) {
  free(s->table);
  s->table = NULL;
}

/* set_insert_clean: table has no dummies and does not contain key */
static inline void pyset_insert_clean(int32_t *table, int64_t mask, int32_t key) {
  uint64_t perturb = (uint64_t)key;
  uint64_t i = (uint64_t)key & (uint64_t)mask;
  for (;;) {
    int32_t *e = &table[i];
    int probes = (i + PYSET_LINEAR_PROBES <= (uint64_t)mask) ? PYSET_LINEAR_PROBES : 0;
    do {
      if (*e == PYSET_EMPTY) {
        *e = key;
        return;
      }
      e++;
    } while (probes--);
    perturb >>= PYSET_PERTURB_SHIFT;
    i = (i * 5 + 1 + perturb) & (uint64_t)mask;
  }
}

static inline void pyset_resize(pyset *s, int64_t minused) {
  int64_t newsize = PYSET_MINSIZE;
  while (newsize <= minused) newsize <<= 1;
  int32_t *old = s->table;
  int64_t oldmask = s->mask;
  int32_t *nt = (int32_t *)malloc(sizeof(int32_t) * (size_t)newsize);
  for (int64_t i = 0; i < newsize; i++) nt[i] = PYSET_EMPTY;
  for (int64_t i = 0; i <= oldmask; i++)
    if (old[i] >= 0) pyset_insert_clean(nt, newsize - 1, old[i]);
  free(old);
  s->table = nt;
  s->mask = newsize - 1;
  s->fill = s->used;
}

/* set_add_entry */
static inline void pyset_add(pyset *s, int32_t key) {
  uint64_t mask = (uint64_t)s->mask;
  uint64_t i = (uint64_t)key & mask;
  uint64_t perturb = (uint64_t)key;
  int32_t *freeslot = NULL;
  int32_t *e;
  for (;;) {
    e = &s->table[i];
    int probes = (i + PYSET_LINEAR_PROBES <= mask) ? PYSET_LINEAR_PROBES : 0;
    do {
      if (*e == PYSET_EMPTY) goto found_unused_or_dummy;
      if (*e == key) return; /* already present */
      if (*e == PYSET_DUMMY) freeslot = e;
      e++;
    } while (probes--);
    perturb >>= PYSET_PERTURB_SHIFT;
    i = (i * 5 + 1 + perturb) & mask;
  }
found_unused_or_dummy:
  if (freeslot) {
    s->used++;
    *freeslot = key;
    return;
  }
  s->fill++;
  s->used++;
  *e = key;
  if ((uint64_t)s->fill * 5 < mask * 3) return;
  pyset_resize(s, s->used > 50000 ? s->used * 2 : s->used * 4);
}

/* set_lookkey + discard; returns 1 if it was present */
static inline int pyset_remove(pyset *s, int32_t key) {
  uint64_t mask = (uint64_t)s->mask;
  uint64_t i = (uint64_t)key & mask;
  uint64_t perturb = (uint64_t)key;
  for (;;) {
    int32_t *e = &s->table[i];
    int probes = (i + PYSET_LINEAR_PROBES <= mask) ? PYSET_LINEAR_PROBES : 0;
    do {
      if (*e == PYSET_EMPTY) return 0;
      if (*e == key) {
        *e = PYSET_DUMMY;
        s->used--;
        return 1;
      }
      e++;
    } while (probes--);
    perturb >>= PYSET_PERTURB_SHIFT;
    i = (i * 5 + 1 + perturb) & mask;
  }
}

static inline int pyset_contains(const pyset *s, int32_t key) {
  uint64_t mask = (uint64_t)s->mask;
  uint64_t i = (uint64_t)key & mask;
  uint64_t perturb = (uint64_t)key;
  for (;;) {
    const int32_t *e = &s->table[i];
    int probes = (i + PYSET_LINEAR_PROBES <= mask) ? PYSET_LINEAR_PROBES : 0;
    do {
      if (*e == PYSET_EMPTY) return 0;
      if (*e == key) return 1;
      e++;
    } while (probes--);
    perturb >>= PYSET_PERTURB_SHIFT;
    i = (i * 5 + 1 + perturb) & mask;
  }
}

/* set_pop: requires used > 0 */
static inline int32_t pyset_pop(pyset *s) {
  uint64_t mask = (uint64_t)s->mask;
  uint64_t i = (uint64_t)s->finger & mask;
  while (s->table[i] < 0) {
    i++;
    if (i > mask) i = 0;
  }
  int32_t key = s->table[i];
  s->table[i] = PYSET_DUMMY;
  s->used--;
  s->finger = (int64_t)(i + 1);
  return key;
}

/* set.copy() / set(other_set): set_merge into a fresh empty set */
static inline void pyset_copy(pyset *dst, const pyset *src) {
  pyset_init(dst);
  if (src->used == 0) return;
  if ((dst->fill + src->used) * 5 >= dst->mask * 3) pyset_resize(dst, (dst->used + src->used) * 2);
  if (dst->mask == src->mask && src->fill == src->used) {
    memcpy(dst->table, src->table, sizeof(int32_t) * (size_t)(src->mask + 1));
    dst->fill = src->fill;
    dst->used = src->used;
    return;
  }
  dst->fill = src->used;
  dst->used = src->used;
  for (int64_t i = 0; i <= src->mask; i++)
    if (src->table[i] >= 0) pyset_insert_clean(dst->table, dst->mask, src->table[i]);
}

/* iteration order: ascending slot index. Returns count, writes keys to out (cap >= used). */
static inline int64_t pyset_list(const pyset *s, int32_t *out) {
  int64_t n = 0;
  for (int64_t i = 0; i <= s->mask; i++)
    if (s->table[i] >= 0) out[n++] = s->table[i];
  return n;
}

#endif
