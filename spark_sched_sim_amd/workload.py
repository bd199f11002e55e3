"""Synthetic TPC-H-format workload + the flat "workload pack" the HIP path reads.

The reference draws every job from 22 queries x 7 input sizes of a downloaded
TPC-H trace set (reference `spark_sched_sim/data_samplers/tpch.py:13-15,117-132`):
per (size, query) one `adj_mat_<q>.npy` (stage DAG) and one pickled
`task_duration_<q>.npy` dict `stage -> wave -> executor level -> [durations]`.
That download is not reproducible offline, so this module

  * generates a *synthetic* trace set with the same schema from a frozen seed
    (`make_raw_workload`),
  * can write it out in the reference's on-disk layout (`write_reference_layout`)
    so that the reference itself can be run on it when fixtures are (re)generated,
  * compiles it into one flat, pointer-free binary blob (`build_pack`) that the
    C oracle and the HIP kernels both consume.

Everything that the reference computes per job at `reset()` but that is in fact a
constant of the (query, size) template is folded into the pack:

  * `num_tasks` from the *pre-cleaning* first key (tpch.py:185-187),
  * the multiset "cleaning" of first_wave and nearest-lower-level fill
    (tpch.py:134-159),
  * the rough mean duration over all three waves after cleaning, duplicates from
    the fill included (tpch.py:161-174),
  * parent/child sets and the row-major `(u, v)` edge list networkx yields for
    `from_numpy_array(adj, DiGraph)` (tpch.py:199; job.py:76-79).
"""
from __future__ import annotations

import hashlib
import os
import os.path as osp
import struct
from typing import Any

import numpy as np

QUERY_SIZES = ["2g", "5g", "10g", "20g", "50g", "80g", "100g"]  # tpch.py:14
NUM_QUERIES = 22  # tpch.py:15
NUM_TEMPLATES = NUM_QUERIES * len(QUERY_SIZES)
WAVES = ("fresh_durations", "first_wave", "rest_wave")
W_FRESH, W_FIRST, W_REST = 0, 1, 2

DEFAULT_SEED = 20240607
PACK_MAGIC = b"SSSPACK2"

_SIZE_SCALE = {"2g": 1, "5g": 2, "10g": 4, "20g": 8, "50g": 20, "80g": 32, "100g": 40}
_LEVELS = [2, 5, 10, 20, 40, 50, 60, 80, 100]

# Trace-set regimes of the synthetic generator (`make_raw_workload(profile=...)`). "default" is the frozen set every round-1..5
# fixture was recorded on (2-18 stages, parents among the 6 nearest predecessors, <= ~220 tasks per stage, 2.7 MB
# pack). "deep" is shaped like what a real TPC-H trace set is expected to look like: DAGs of up to 40 stages whose parents sit
# anywhere upstream (in-degree <= 6), 4 .. 3000 tasks per stage growing with the input size, the executor levels the sampler's
# interval table actually produces (tpch.py:238: no level 2), task durations from 50 ms to 40 s - a pack of tens of MB (past the
# aggregate L2) with long runs of task completions per scheduling decision. Same corner-case rotation as the default.
PROFILES = {
    "default": dict(stages=(2, 19), max_in=3, parent_window=6, tasks=(1, 12), tasks_div=2, base=(200, 4000), levels=_LEVELS),
    "deep": dict(stages=(4, 41), max_in=6, parent_window=None, tasks=(4, 76), tasks_div=1, base=(50, 20000), levels=_LEVELS[1:]),
}


def template_index(query_num: int, size_idx: int, n_sizes: int = len(QUERY_SIZES)) -> int:
    """template id used by the pack: query-major. `query_num` is 1-based as in
    tpch.py:177, `size_idx` indexes QUERY_SIZES as drawn in tpch.py:178."""
    return (query_num - 1) * n_sizes + size_idx


def trace_set_shape(raw: dict) -> tuple[list[str], int]:
    """(query sizes, number of queries) of a raw trace set {(size, query_num): ...}: the reference's constants QUERY_SIZES /
    NUM_QUERIES (tpch.py:14-15) for its own dataset; a user's trace set may have any number of either. Sizes in the reference's
    order where they are the reference's, else as first met; queries must be numbered 1..n for every size."""
    sizes: list[str] = []
    for size, _ in raw:
        if size not in sizes:
            sizes.append(size)
    if set(sizes) <= set(QUERY_SIZES):
        sizes.sort(key=QUERY_SIZES.index)
    n_q = max(q for _, q in raw)
    missing = [(sz, q) for sz in sizes for q in range(1, n_q + 1) if (sz, q) not in raw]
    if missing:
        raise ValueError(f"trace set is not a full (size x query) grid: missing {missing[:4]}")
    return sizes, n_q


# --------------------------------------------------------------------------
# raw synthetic trace set (reference schema)
# --------------------------------------------------------------------------


def _make_dag(rng: np.random.Generator, n: int, max_in: int = 3, parent_window: int | None = 6) -> np.ndarray:
    """random DAG on n >= 2 stages, edges u < v, in-degree <= max_in (parents among the `parent_window` nearest
    predecessors; None: any predecessor), >= 1 edge (the reference's `_reset_edge_links` cannot handle an edge-less job,
    spark_sched_sim.py:254)."""
    adj = np.zeros((n, n), dtype=np.int64)
    for v in range(1, n):
        max_par = min(v, max_in)
        # sources are allowed (k = 0) but get rarer further down the DAG
        k = int(rng.integers(0, max_par + 1))
        if k == 0 and rng.random() < 0.7:
            k = 1
        if k:
            lo = 0 if parent_window is None else max(0, v - parent_window)
            cand = np.arange(lo, v)
            k = min(k, cand.size)
            par = rng.choice(cand, size=k, replace=False)
            adj[par, v] = 1
    if adj.sum() == 0:
        adj[0, n - 1] = 1
    return adj


def _make_stage_durations(
    rng: np.random.Generator, scale: int, variant: int, prof: dict | None = None
) -> dict[str, dict[int, list[int]]]:
    """one stage's `{wave: {level: [ms, ...]}}` dict.

    `variant` rotates through corner cases so that every branch of
    `TPCHDataSampler.task_duration` (tpch.py:75-106) and of the cleaning pass
    (tpch.py:134-159) is hit by some template:
      1 -> highest levels missing from first_wave (=> `max(first_wave)` path)
      2 -> rest_wave misses some levels (KeyError fallback)
      3 -> a mid level whose first_wave is exactly its fresh list
           (cleaning empties it => inherits the lower level's list)
      4 -> empty fresh list at some level (ValueError fallback + warmup)
      5 -> keys inserted in descending order (first key != smallest)
    """
    prof = PROFILES["default"] if prof is None else prof
    num_tasks = max(1, int(rng.integers(*prof["tasks"])) * scale // prof["tasks_div"])
    base = int(rng.integers(*prof["base"]))

    levels = list(prof["levels"])
    if variant == 1:
        levels = levels[: int(rng.integers(3, 6))]
    if variant == 5:
        levels = levels[::-1]

    first: dict[int, list[int]] = {}
    rest: dict[int, list[int]] = {}
    fresh: dict[int, list[int]] = {}
    for e in levels:
        n_first = max(1, min(num_tasks, e))
        n_rest = num_tasks - n_first
        fw = rng.integers(base, 2 * base, size=n_first).tolist()
        rw = rng.integers(max(1, base // 2), base, size=n_rest).tolist()
        # fresh durations: a prefix of the first wave (these get removed from
        # first_wave by the cleaning pass) plus a few values of their own
        k = (n_first + 1) // 2
        fr = fw[:k] if n_first > 1 else []
        fr = fr + (rng.integers(base, 2 * base, size=max(1, k // 2)) + base).tolist()
        first[e], rest[e], fresh[e] = fw, rw, fr

    lv_sorted = sorted(levels)
    if variant == 2 and len(lv_sorted) > 2:
        for e in lv_sorted[1:]:
            if rng.random() < 0.5:
                del rest[e]
    if variant == 3 and len(lv_sorted) > 3:
        e = lv_sorted[int(rng.integers(1, 4))]
        fresh[e] = list(first[e])
    if variant == 4:
        e = lv_sorted[int(rng.integers(0, min(4, len(lv_sorted))))]
        fresh[e] = []

    return {"fresh_durations": fresh, "first_wave": first, "rest_wave": rest}


def make_raw_workload(seed: int = DEFAULT_SEED, query_sizes: list[str] | None = None, num_queries: int = NUM_QUERIES,
                      profile: str = "default") -> dict[tuple[str, int], tuple[np.ndarray, dict]]:
    """-> {(size, query_num): (adj_mat int64[S,S], {stage: {wave: {level: [int]}}})}; by default the reference's grid of
    7 sizes x 22 queries (the frozen set the fixtures were recorded on); other grids for tests of other trace-set shapes;
    `profile`: the regime of DAG shapes / task counts / durations (PROFILES)"""
    prof = PROFILES[profile] if isinstance(profile, str) else dict(profile)  # (a dict: generator parameters of one's own, e.g. the tests' random regimes)
    rng = np.random.default_rng(seed)
    raw: dict[tuple[str, int], tuple[np.ndarray, dict]] = {}
    query_sizes = list(QUERY_SIZES) if query_sizes is None else list(query_sizes)
    for q in range(1, num_queries + 1):
        n_stages = int(rng.integers(*prof["stages"]))
        adj = _make_dag(rng, n_stages, prof["max_in"], prof["parent_window"])
        for size in query_sizes:
            scale = _SIZE_SCALE[size]
            td = {}
            for s in range(n_stages):
                variant = int(rng.integers(0, 8))
                td[s] = _make_stage_durations(rng, scale, variant, prof)
            raw[(size, q)] = (adj.copy(), td)
    return raw


def write_reference_layout(raw: dict, root: str) -> None:
    """writes `<root>/data/tpch/<size>/{adj_mat,task_duration}_<q>.npy`, the layout
    `TPCHDataSampler._load_query` reads relative to cwd (tpch.py:117-132)."""
    for (size, q), (adj, td) in raw.items():
        d = osp.join(root, "data", "tpch", size)
        os.makedirs(d, exist_ok=True)
        np.save(osp.join(d, f"adj_mat_{q}.npy"), adj, allow_pickle=True)
        np.save(osp.join(d, f"task_duration_{q}.npy"), np.array(td, dtype=object), allow_pickle=True)


# --------------------------------------------------------------------------
# template constants the reference recomputes per job
# --------------------------------------------------------------------------


def clean_first_wave(td: dict[str, dict[int, list[int]]]) -> dict[int, list[int]]:
    """multiset-subtract the fresh durations from first_wave per level, then let an
    emptied level inherit the nearest lower level's list (tpch.py:134-159)."""
    clean: dict[int, list[int]] = {}
    for e, fw in td["first_wave"].items():
        budget: dict[int, int] = {}
        for d in td["fresh_durations"][e]:
            budget[d] = budget.get(d, 0) + 1
        kept = []
        for d in fw:
            if budget.get(d, 0) > 0:
                budget[d] -= 1
            else:
                kept.append(d)
        clean[e] = kept
    last: list[int] = []
    for e in sorted(clean):
        if not clean[e]:
            clean[e] = last
        last = clean[e]
    return clean


def _template_constants(adj: np.ndarray, td: dict) -> dict[str, Any]:
    n = adj.shape[0]
    assert adj.shape == (n, n) and len(td) == n
    stages = []
    for s in range(n):
        data = td[s]
        e0 = next(iter(data["first_wave"]))
        num_tasks = len(data["first_wave"][e0]) + len(data["rest_wave"][e0])  # tpch.py:185-187
        first = clean_first_wave(data)
        allv = (
            [t for ts in data["fresh_durations"].values() for t in ts]
            + [t for ts in first.values() for t in ts]
            + [t for ts in data["rest_wave"].values() for t in ts]
        )
        rough = float(np.mean(allv))  # tpch.py:161-174
        stages.append(
            {
                "num_tasks": num_tasks,
                "rough": rough,
                "waves": (data["fresh_durations"], first, data["rest_wave"]),
            }
        )
    us, vs = np.nonzero(adj)  # row-major == networkx edge order for from_numpy_array
    return {"n": n, "stages": stages, "edges": list(zip(us.tolist(), vs.tolist()))}


def _is_dag(n: int, edges: list[tuple[int, int]]) -> bool:
    indeg = [0] * n
    for _, v in edges:
        indeg[v] += 1
    todo = [v for v in range(n) if indeg[v] == 0]
    seen = 0
    while todo:
        u = todo.pop()
        seen += 1
        for a, b in edges:
            if a == u:
                indeg[b] -= 1
                if indeg[b] == 0:
                    todo.append(b)
    return seen == n


# --------------------------------------------------------------------------
# pack
# --------------------------------------------------------------------------

_SECTIONS = (
    ("levels", np.int32),
    ("tmpl_stage_off", np.int32),
    ("tmpl_edge_off", np.int32),
    ("stage_num_tasks", np.int32),
    ("stage_rough", np.float64),
    ("stage_parent_mask", np.uint64),
    ("stage_child_mask", np.uint64),
    ("stage_first_keymask", np.uint32),
    ("stage_max_first_lvl", np.int32),
    ("edges", np.int32),
    ("desc", np.int32),
    ("durations", np.int32),
)


MAX_STAGES_PER_JOB = 64       # stage bit masks are 64 bits (csrc/sss_layout.h: SSS_MAX_STAGES)
MAX_EDGES_PER_JOB = 255       # SssJob.n_edges is a byte
MAX_LEVELS = 16               # csrc/sss_layout.h: SSS_MAX_LEVELS
MAX_INT32 = 2 ** 31 - 1       # task counts, durations (ms), list offsets and lengths are carried as int32


def _duration_array(lst, where: str) -> np.ndarray:
    """one duration list as the int32 array the pack carries - or a ValueError naming the list. The reference takes whatever
    numbers the trace files hold (tpch.py:208-214 `np_random.choice(list)`; event times are Python floats); the pack narrows
    them to int32 milliseconds, so everything that does not survive that narrowing unchanged is refused HERE rather than
    truncated: non-integer values (1234.5), values >= 2^31, NaN / inf, negative durations (the batched event paths bound the
    time of events that do not exist yet from below with the lists' minima), non-numeric entries."""
    a = np.asarray(lst)
    if a.ndim != 1:
        raise ValueError(f"{where}: expected a flat list of durations, got shape {a.shape}")
    if a.size == 0:
        return np.zeros(0, dtype=np.int32)
    if a.dtype.kind == "b" or a.dtype.kind not in "iuf":
        raise ValueError(f"{where}: durations must be numbers, got dtype {a.dtype}")
    if a.dtype.kind == "f":
        if not np.all(np.isfinite(a)):
            raise ValueError(f"{where}: non-finite task duration")
        frac = a != np.rint(a)
        if frac.any():
            raise ValueError(f"{where}: task duration {a[frac][0]!r} is not a whole number of milliseconds "
                             "(the pack carries int32 ms; rescale the trace set instead of letting it truncate)")
    if (a < 0).any():
        raise ValueError(f"{where}: negative task duration {a[a < 0][0]!r}")
    if (a > MAX_INT32).any():
        raise ValueError(f"{where}: task duration {a[a > MAX_INT32][0]!r} does not fit int32 milliseconds (max {MAX_INT32})")
    return a.astype(np.int32)


def build_pack_arrays(raw: dict, query_sizes: list[str] | None = None, num_queries: int | None = None) -> dict[str, np.ndarray]:
    """`query_sizes` / `num_queries`: the sampler's QUERY_SIZES / NUM_QUERIES (tpch.py:14-15: `integers(NUM_QUERIES)`,
    `choice(QUERY_SIZES)` index the templates); by default whatever grid `raw` holds (the reference's 7 x 22 for its own data).

    Every field the pack carries narrower than the reference's Python objects is checked here and a ValueError names the
    offending template / stage / list (DESIGN.md section 8 lists the limits); `sss_create` checks what can still be seen in the
    serialized pack again."""
    if query_sizes is None or num_queries is None:
        found_sizes, found_q = trace_set_shape(raw)
        query_sizes = found_sizes if query_sizes is None else list(query_sizes)
        num_queries = found_q if num_queries is None else int(num_queries)
    level_set = set()
    for _, td in raw.values():
        for st in td.values():
            for w in WAVES:
                level_set.update(st[w].keys())
    levels = sorted(level_set)
    L = len(levels)
    if L > MAX_LEVELS:
        raise ValueError(f"trace set uses {L} distinct executor levels {levels}; at most {MAX_LEVELS} are supported")
    if any((not isinstance(e, (int, np.integer))) or isinstance(e, bool) or e < 0 or e > MAX_INT32 for e in levels):
        raise ValueError(f"executor levels must be non-negative ints, got {levels}")
    lvl_idx = {e: i for i, e in enumerate(levels)}

    tmpl_stage_off = [0]
    tmpl_edge_off = [0]
    num_tasks, rough, pmask, cmask, keymask, maxlvl = [], [], [], [], [], []
    edges: list[tuple[int, int]] = []
    desc_rows = []
    dur_chunks: list[np.ndarray] = []
    dur_off = 0

    for q in range(1, num_queries + 1):
        for size in query_sizes:
            adj, td = raw[(size, q)]
            name = f"template (size {size!r}, query {q})"
            adj = np.asarray(adj)
            n = adj.shape[0] if adj.ndim == 2 else -1
            if adj.ndim != 2 or adj.shape != (n, n) or len(td) != n:
                raise ValueError(f"{name}: adjacency matrix {adj.shape} does not match its {len(td)} stages")
            if not 1 <= n <= MAX_STAGES_PER_JOB:
                raise ValueError(f"{name}: {n} stages; 1 .. {MAX_STAGES_PER_JOB} per job are supported (stage bit masks are 64 bits)")
            for s_, st_ in td.items():  # (before anything is computed from the lists: the cleaning pass and the rough mean)
                for w_ in WAVES:
                    for e_, lst_ in st_[w_].items():
                        _duration_array(lst_, f"{name} stage {s_} {w_}[{e_}]")
            tc = _template_constants(adj, td)
            if not tc["edges"]:
                raise ValueError(f"{name}: no edge - the reference cannot run an edge-less job either (spark_sched_sim.py:254 np.vstack([]))")
            if len(tc["edges"]) > MAX_EDGES_PER_JOB:
                raise ValueError(f"{name}: {len(tc['edges'])} edges; at most {MAX_EDGES_PER_JOB} per job are supported")
            if any(u >= v for u, v in tc["edges"]) and not _is_dag(n, tc["edges"]):
                raise ValueError(f"{name}: the stage graph has a cycle")
            pm = [0] * n
            cm = [0] * n
            for u, v in tc["edges"]:
                pm[v] |= 1 << u
                cm[u] |= 1 << v
            edges += tc["edges"]
            for s, st in enumerate(tc["stages"]):
                if not 0 <= st["num_tasks"] <= MAX_INT32:
                    raise ValueError(f"{name} stage {s}: {st['num_tasks']} tasks do not fit the int32 task counter")
                num_tasks.append(st["num_tasks"])
                rough.append(st["rough"])
                pmask.append(pm[s])
                cmask.append(cm[s])
                fresh, first, rest = st["waves"]
                km = 0
                for e in first:
                    km |= 1 << lvl_idx[e]
                keymask.append(km)
                maxlvl.append(lvl_idx[max(first)])
                d = np.full((3, L, 2), -1, dtype=np.int32)
                d[:, :, 0] = 0
                seen: dict[int, tuple[int, int]] = {}
                for w, wave in enumerate((fresh, first, rest)):
                    for e, lst in wave.items():
                        key = id(lst)
                        if key not in seen:  # inherited lists share storage
                            arr = _duration_array(lst, f"{name} stage {s} {WAVES[w]}[{e}]")
                            if arr.size >= 1 << 30:
                                raise ValueError(f"{name} stage {s} {WAVES[w]}[{e}]: {arr.size} durations in one list (max 2^30 - 1)")
                            seen[key] = (dur_off, arr.size)
                            dur_chunks.append(arr)
                            dur_off += arr.size
                            if dur_off > MAX_INT32:
                                raise ValueError(f"trace set holds more than {MAX_INT32} durations (list offsets are int32)")
                        d[w, lvl_idx[e]] = seen[key]
                desc_rows.append(d)
            tmpl_stage_off.append(len(num_tasks))
            tmpl_edge_off.append(len(edges))

    return {
        "n_sizes": len(query_sizes),
        "levels": np.asarray(levels, dtype=np.int32),
        "tmpl_stage_off": np.asarray(tmpl_stage_off, dtype=np.int32),
        "tmpl_edge_off": np.asarray(tmpl_edge_off, dtype=np.int32),
        "stage_num_tasks": np.asarray(num_tasks, dtype=np.int32),
        "stage_rough": np.asarray(rough, dtype=np.float64),
        "stage_parent_mask": np.asarray(pmask, dtype=np.uint64),
        "stage_child_mask": np.asarray(cmask, dtype=np.uint64),
        "stage_first_keymask": np.asarray(keymask, dtype=np.uint32),
        "stage_max_first_lvl": np.asarray(maxlvl, dtype=np.int32),
        "edges": np.asarray(edges, dtype=np.int32).reshape(-1, 2),
        "desc": np.stack(desc_rows).astype(np.int32),
        "durations": np.concatenate(dur_chunks).astype(np.int32),
    }


def serialize_pack(arrs: dict[str, np.ndarray]) -> bytes:
    """flat little-endian blob:
    magic[8] | i64 header[8] | i64 toc[nsec][2] (byte offset, byte length) | sections (8-aligned)

    header = (n_templates, n_levels, s_max, total_stages, total_edges,
              total_durations, n_sections, n_sizes or 0)

    n_templates = n_queries * n_sizes (a job's template is query * n_sizes + size, tpch.py:177-178). The last word holds n_sizes,
    or 0 for the reference's seven: packs of 7-size trace sets - the frozen default, whose digest the fixtures record - stay
    byte for byte what they were before the word had a meaning.
    """
    T = arrs["tmpl_stage_off"].size - 1
    L = arrs["levels"].size
    per_t = np.diff(arrs["tmpl_stage_off"])
    header = [
        T,
        L,
        int(per_t.max()),
        int(arrs["stage_num_tasks"].size),
        int(arrs["edges"].shape[0]),
        int(arrs["durations"].size),
        len(_SECTIONS),
        0 if int(arrs.get("n_sizes", 7)) == len(QUERY_SIZES) else int(arrs["n_sizes"]),
    ]
    assert T % int(arrs.get("n_sizes", 7)) == 0
    head_bytes = 8 + 8 * len(header) + 16 * len(_SECTIONS)
    off = (head_bytes + 7) & ~7
    toc = []
    blobs = []
    for name, dt in _SECTIONS:
        b = np.ascontiguousarray(arrs[name], dtype=dt).tobytes()
        toc.append((off, len(b)))
        pad = (-len(b)) & 7
        blobs.append(b + b"\0" * pad)
        off += len(b) + pad
    out = bytearray()
    out += PACK_MAGIC
    out += struct.pack("<8q", *header)
    for o, n in toc:
        out += struct.pack("<2q", o, n)
    out += b"\0" * (((head_bytes + 7) & ~7) - head_bytes)
    for b in blobs:
        out += b
    return bytes(out)


def pack_section(pack: bytes, name: str) -> np.ndarray:
    """numpy view of one section of a serialized pack"""
    names = [n for n, _ in _SECTIONS]
    i = names.index(name)
    head = 8 + 8 * 8
    off, length = struct.unpack_from("<2q", pack, head + 16 * i)
    return np.frombuffer(pack, dtype=_SECTIONS[i][1], count=length // np.dtype(_SECTIONS[i][1]).itemsize, offset=off)


def pack_max_depth(pack: bytes) -> int:
    """the longest path (in edges) of any template's stage DAG: an upper bound of the number of DAG layers
    (topological generations beyond the first, decima/utils.py:246-267) an observation of this workload can have"""
    edges, e_off, s_off = pack_section(pack, "edges"), pack_section(pack, "tmpl_edge_off"), pack_section(pack, "tmpl_stage_off")
    best = 0
    for t in range(len(e_off) - 1):
        n = int(s_off[t + 1] - s_off[t])
        es = np.asarray(edges).reshape(-1, 2)[int(e_off[t]): int(e_off[t + 1])].tolist()
        dist = [0] * n
        for _ in range(n):  # longest-path relaxation (the DAGs have at most 64 stages)
            moved = False
            for u, v in es:
                if dist[v] < dist[u] + 1:
                    dist[v], moved = dist[u] + 1, True
            if not moved:
                break
        best = max(best, max(dist, default=0))
    return best


def build_pack(raw: dict | None = None, seed: int = DEFAULT_SEED, query_sizes: list[str] | None = None, num_queries: int | None = None) -> bytes:
    if raw is None:
        raw = make_raw_workload(seed)
    return serialize_pack(build_pack_arrays(raw, query_sizes, num_queries))


def pack_shape(pack: bytes) -> tuple[int, int]:
    """(number of queries, number of sizes) of a serialized pack"""
    h = struct.unpack_from("<8q", pack, 8)
    n_sizes = h[7] if h[7] > 0 else len(QUERY_SIZES)
    return h[0] // n_sizes, n_sizes


def pack_digest(pack: bytes) -> str:
    return hashlib.sha256(pack).hexdigest()


_CACHE: dict = {}


def default_pack(seed: int = DEFAULT_SEED) -> bytes:
    """the frozen synthetic pack (cached per process)."""
    if seed not in _CACHE:
        _CACHE[seed] = build_pack(seed=seed)
    return _CACHE[seed]


def profile_pack(profile: str = "default", seed: int = DEFAULT_SEED, query_sizes: list[str] | None = None, num_queries: int = NUM_QUERIES) -> bytes:
    """the pack of the synthetic trace set of a generator profile (PROFILES), cached per process"""
    key = (profile, seed, tuple(query_sizes) if query_sizes is not None else None, num_queries)
    if key not in _CACHE:
        _CACHE[key] = default_pack(seed) if key == ("default", seed, None, NUM_QUERIES) else build_pack(make_raw_workload(seed, query_sizes, num_queries, profile=profile))
    return _CACHE[key]


def pack_from_reference_layout(root: str, query_sizes: list[str] | None = None, num_queries: int | None = None) -> bytes:
    """compile a trace set stored in the reference's on-disk layout (e.g. the real
    TPC-H traces, if a user has them) into a pack. `query_sizes` / `num_queries` default to the reference's constants
    (tpch.py:14-15: 7 sizes x 22 queries); a user whose trace set differs passes theirs, as they would edit those two lines."""
    query_sizes = list(QUERY_SIZES) if query_sizes is None else list(query_sizes)
    num_queries = NUM_QUERIES if num_queries is None else int(num_queries)
    raw = {}
    for size in query_sizes:
        for q in range(1, num_queries + 1):
            d = osp.join(root, "data", "tpch", size)
            adj = np.load(osp.join(d, f"adj_mat_{q}.npy"), allow_pickle=True)
            td = np.load(osp.join(d, f"task_duration_{q}.npy"), allow_pickle=True).item()
            raw[(size, q)] = (adj, td)
    return build_pack(raw, query_sizes=query_sizes, num_queries=num_queries)
