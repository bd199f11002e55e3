"""Test builds of the HIP library (TEST INFRASTRUCTURE): the product sources compiled with an extra -D into tests/_build/, for
comparisons between formulations of the same kernel. The product library (spark_sched_sim_amd/csrc/libsss_hip.so) has no
run-time switches; `-DSSS_TEST_VECTOR_FORMS` selects the vector-unit GNN / MLP kernels (csrc/sss_gnn.h, sss_gnn16.h,
sss_train16.h) where the product takes the matrix-core ones (csrc/sss_hip.hip: kVectorForms)."""
import ctypes
import os
import os.path as osp

HERE = osp.dirname(osp.abspath(__file__))
VARIANTS = {"vecforms": ("-DSSS_TEST_VECTOR_FORMS",),
            # the one-launch DAG layers (csrc/sss_gnn_mfma.h) with room for 64 list entries per observation: larger observations take
            # the chunk-by-chunk path
            "obscap64": ("-DGNN_OBS_LIST_CAP=64",),
            # timing builds for tools/debug/evprof3.py (scoped profiler of the lane-0 procedures, csrc/sss_prof.h)
            "evprof3": ("-DSSS_EVPROF3",), "evprof3b": ("-DSSS_EVPROF3", "-DSSS_EVPROF3B"), "evprof3c": ("-DSSS_EVPROF3", "-DSSS_EVPROF3C"),
            "evprof3d": ("-DSSS_EVPROF3", "-DSSS_EVPROF3D"),
            # batch-size thresholds of the event loop (A/B timing: `bench.py --lib tests/_build/libsss_hip_<name>.so`)
            "thr22": ("-DSSS_MIN_RELEASED_BATCH=2", "-DSSS_MIN_ARRIVAL_BATCH=2"), "thr32": ("-DSSS_MIN_RELEASED_BATCH=3", "-DSSS_MIN_ARRIVAL_BATCH=2"),
            "thr63": ("-DSSS_MIN_RELEASED_BATCH=6", "-DSSS_MIN_ARRIVAL_BATCH=3"), "arr1": ("-DSSS_MIN_ARRIVAL_BATCH=1",),
            # timing only (wrong durations): the fast run without its load from the duration pool - what that load costs per event
            "noload": ("-DSSS_EXP_NOLOAD",),
            # A/B timing: which lanes of a fast run load their candidate duration (csrc/sss_sim_fast_run.h SSS_EXP_DUR)
            "loadpred": ("-DSSS_FAST_LOAD_PRED",), "loadsel": ("-DSSS_FAST_LOAD_SEL",), "loadall": ("-DSSS_FAST_LOAD_ALL",),
            "nolean": ("-DSSS_NO_LEAN",), "syncwg": ("-DSSS_SYNC_WORKGROUP",), "pair17": ("-DSSS_PAIR_MIN_E=17",), "w3": ("-DSSS_WAVES_PER_SIMD=3",), "w2": ("-DSSS_WAVES_PER_SIMD=2",), "oplane": ("-DSSS_OPAQUE_LANE",), "mlicm": ("-mllvm", "-disable-machine-licm=false"),
            # csrc/sss_rows.h: the atomic additions with 16 bytes per lane (A/B timing, tools/debug/rows_time.py)
            "vecatom": ("-DSSS_ROWS_VEC_ATOMICS=1",),
            # job-cache slots at large job capacities (csrc/sss_layout.h; A/B timing at small env counts, profiles/r04_bench.md section 9)
            "slots6": ("-DSSS_FALLBACK_SLOTS=6",), "slots8": ("-DSSS_FALLBACK_SLOTS=8",),
            "slots16": ("-DSSS_FALLBACK_SLOTS=16",), "slots24": ("-DSSS_FALLBACK_SLOTS=24",), "slots32": ("-DSSS_FALLBACK_SLOTS=32",), "slots48": ("-DSSS_FALLBACK_SLOTS=48",)}


def variant_path(name: str) -> str:
    return osp.join(HERE, "_build", f"libsss_hip_{name}.so")


def build_variant(name: str, verbose: bool = False) -> str:
    """builds (if stale) and returns the path; hipcc cross-compiles, so the build container can do it ahead of the GPU box"""
    from spark_sched_sim_amd import build as hip_build

    os.makedirs(osp.join(HERE, "_build"), exist_ok=True)
    return hip_build.build(out=variant_path(name), extra_flags=VARIANTS[name], verbose=verbose)


def load_variant(name: str) -> ctypes.CDLL:
    return ctypes.CDLL(build_variant(name))


if __name__ == "__main__":  # python tests/gpu_variant.py evprof3 ...: build ahead of a gpurun call (the .so travels with the snapshot)
    import sys

    sys.path.insert(0, osp.dirname(HERE))
    for n in sys.argv[1:]:
        print(build_variant(n, verbose=False))
