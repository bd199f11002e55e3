"""No-GPU checks of the product library: it builds for gfx950, loads, exports every symbol that
include/sss.h declares, validates its inputs host-side, and the Python host refuses to run
without the HIP path (no CPU fallback)."""
import ctypes as C
import os.path as osp
import re
import subprocess

import numpy as np
import pytest

ROOT = osp.dirname(osp.dirname(osp.abspath(__file__)))


@pytest.fixture(scope="module")
def hip_lib():
    from spark_sched_sim_amd import build

    return C.CDLL(build.build())


def test_exports_match_header(hip_lib):
    header = open(osp.join(ROOT, "include", "sss.h")).read()
    declared = set(re.findall(r"\b(sss_[a-z_]+)\s*\(", header))
    from spark_sched_sim_amd.binding import EXPORTS

    assert declared == set(EXPORTS)
    for sym in declared:
        assert hasattr(hip_lib, sym), sym


def test_binding_mirrors_have_the_library_sizes(hip_lib):
    """every ctypes mirror of an argument structure (binding.py) is as large as the structure the library was compiled with
    (include/sss.h sss_abi_sizeof) and as the header compiled with gcc says - a stale binding fails at load time (Binding.check_abi),
    not inside a kernel (round 5 grew sss_decima_graph's list counters from i64[32] to i64[33][32] without any such guard)"""
    from spark_sched_sim_amd.binding import ABI_STRUCTS, Binding

    hip_lib.sss_abi_sizeof.argtypes = [C.c_char_p]
    header = open(osp.join(ROOT, "include", "sss.h")).read()
    assert set(re.findall(r"^} (sss_[a-z_]+);", header, re.M)) == set(ABI_STRUCTS)   # every structure of the header has a mirror
    for name, cls in ABI_STRUCTS.items():
        assert hip_lib.sss_abi_sizeof(name.encode()) == C.sizeof(cls), name
    assert hip_lib.sss_abi_sizeof(b"no_such_struct") == -1
    Binding(hip_lib)   # (check_abi runs in the constructor)

    class Stale(C.Structure):   # the round-5 layout of sss_decima_graph: one field short
        _fields_ = ABI_STRUCTS["sss_decima_graph"]._fields_[:-1]

    import spark_sched_sim_amd.binding as bnd
    keep = bnd.ABI_STRUCTS["sss_decima_graph"]
    bnd.ABI_STRUCTS["sss_decima_graph"] = Stale
    try:
        with pytest.raises(RuntimeError, match="binding / library mismatch"):
            Binding(hip_lib)
    finally:
        bnd.ABI_STRUCTS["sss_decima_graph"] = keep


def test_simulator_kernels_do_not_spill(hip_lib):
    """The simulator units are compiled with machine LICM off (spark_sched_sim_amd/build.py: with the pass on, 32 loop-invariant
    register pairs are hoisted out of the event loop and spilled across it - 272 / 640 bytes of scratch per lane in the step /
    rollout kernels, 16 KB of spill stores per env-step). That is a compiler-internal switch: should a toolchain stop honouring it,
    this fails instead of the spills silently coming back. Scratch per lane and vector-register spills of the built library's
    kernels, read from its code objects (tools/isa_counts.py kernel_metadata); the 4-waves-per-SIMD register budget with them."""
    import sys

    from spark_sched_sim_amd import build

    sys.path.insert(0, osp.join(ROOT, "tools"))
    from isa_counts import kernel_metadata

    md = kernel_metadata(build.build())
    for kernel, scratch_max in (("sss_step_kernel", 64), ("sss_step_bounded_kernel", 64), ("sss_rollout_kernel", 96), ("sss_reset_kernel", 0),
                                ("sss_step_kernel_wide", 64), ("sss_step_bounded_kernel_wide", 64), ("sss_rollout_kernel_wide", 96), ("sss_reset_kernel_wide", 0)):
        k = md[kernel]
        assert k["private_segment_fixed_size"] <= scratch_max, (kernel, k)
        assert k["vgpr_spill_count"] <= 8 and k["vgpr_count"] <= 128, (kernel, k)


def test_matrix_core_kernels_are_in_the_library_without_scratch(hip_lib):
    """The MFMA formulations (csrc/sss_gnn_mfma.h, sss_train16.h) exist only in the gfx950 build and their numerics are -m gpu tests;
    what CAN be held without a GPU is that a toolchain still builds every one of them, really with matrix-core instructions, without
    scratch or register spills, and that the occupancy their launchers count on (registers per lane) has not drifted: read from the
    built library's code objects and its disassembly."""
    import subprocess
    import sys

    from spark_sched_sim_amd import build

    sys.path.insert(0, osp.join(ROOT, "tools"))
    from isa_counts import LLVM_BIN, kernel_metadata

    so = build.build()
    md = kernel_metadata(so)
    want = {"sss_gnn_layer_mfma_kernel": 168, "sss_gnn_rows_mfma_kernel": 128, "sss_gnn_head_mfma_kernel": 128, "sss_mlp_mfma_fwd_kernel": 128,
            "sss_mlp_mfma_bwd_kernel": 128, "sss_mlp_mfma_bwdw_kernel": 256,  # (two waves per SIMD: 152-204 registers)
            "sss_mlp_head_mfma_fwd_kernel": 256, "sss_mlp_head_mfma_bwd_kernel": 168, "sss_mlp_head_mfma_bwdw_kernel": 512}  # (one workgroup per CU: its LDS)
    for stem, vgpr_max in want.items():
        found = [k for k in md if stem in k]
        assert found, stem
        for k in found:
            m = md[k]
            assert m["private_segment_fixed_size"] == 0 and m["vgpr_spill_count"] == 0 and m["sgpr_spill_count"] == 0, (k, m)
            assert m["vgpr_count"] <= vgpr_max, (k, m["vgpr_count"])
    # the recomputing backward kernels (round 6) are there for the three GNN shapes
    assert sum("sss_mlp_mfma_bwdw_kernelILi" in k and "ELb1ELb0E" in k for k in md) == 3
    # ... and the two-piece form of the 21-wide one (forward and backward)
    assert any("sss_mlp_mfma_bwdw_kernelILi21ELb1ELb1E" in k for k in md) and any("sss_mlp_mfma_fwd_kernelILi21ELb1E" in k for k in md)
    # ... and the code really is matrix-core code (a build that silently lost the builtin would fall back to nothing: there is none)
    dis = subprocess.run([osp.join(LLVM_BIN, "llvm-objdump"), "-d", "--offloading", so], capture_output=True, text=True).stdout
    if "v_mfma" not in dis:  # (llvm-objdump of the host library does not show device code: look into the unbundled code objects)
        import glob
        import tempfile
        with tempfile.TemporaryDirectory() as tmp:
            fb = osp.join(tmp, "fatbin")
            subprocess.run([osp.join(LLVM_BIN, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", so, fb], check=True)
            data = open(fb, "rb").read()
            n = 0
            start = 0
            while True:  # every embedded ELF of the fat binary
                at = data.find(b"\x7fELF", start)
                if at < 0:
                    break
                nxt = data.find(b"\x7fELF", at + 4)
                open(osp.join(tmp, f"co{n}.elf"), "wb").write(data[at: nxt if nxt > 0 else len(data)])
                n, start = n + 1, at + 4
            dis = "".join(subprocess.run([osp.join(LLVM_BIN, "llvm-objdump"), "-d", f], capture_output=True, text=True).stdout for f in glob.glob(osp.join(tmp, "co*.elf")))
    assert dis.count("v_mfma_f32_16x16x4_f32") + dis.count("v_mfma_f32_16x16x4f32") > 100, "no fp32 MFMA instructions in the built library"


def test_query_dims_and_validation(hip_lib, pack):
    from spark_sched_sim_amd.binding import Binding, SssCfg

    b = Binding(hip_lib)
    d = b.query_dims(SssCfg(10, 50, 0, 0, 4e-5, 2000.0, 1000.0, 0.0), pack, 4096)
    assert (d.num_envs, d.num_executors, d.job_cap, d.stage_stride) == (4096, 10, 50, 18)
    assert d.node_cap == 50 * 18 and d.env_stride % 256 == 0 and d.state_bytes == d.env_stride * 4096
    for bad_cfg, what in ((SssCfg(129, 50, 0, 0, 4e-5, 2000.0, 1000.0, 0.0), "num_executors"),
                          (SssCfg(10, 0, 0, 0, 4e-5, 2000.0, 1000.0, 0.0), "max_jobs"),
                          (SssCfg(10, 50, 0, 0, 0.0, 2000.0, 1000.0, 0.0), "job_arrival_rate")):
        with pytest.raises(ValueError, match=what):
            b.query_dims(bad_cfg, pack, 8)
    with pytest.raises(ValueError, match="pack"):
        b.query_dims(SssCfg(10, 50, 0, 0, 4e-5, 2000.0, 1000.0, 0.0), b"garbage" * 100, 8)


def test_no_cpu_fallback():
    from spark_sched_sim_amd import VecSparkSchedSimEnv
    from spark_sched_sim_amd.binding import load_library

    cfg = dict(num_executors=10, job_arrival_cap=50, job_arrival_rate=4e-5, moving_delay=2000.0, warmup_delay=1000.0)
    with pytest.raises(RuntimeError, match="no CPU path"):
        VecSparkSchedSimEnv(cfg, 4, device="cpu")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        load_library("/nonexistent/libsss_hip.so")


def test_header_offsets_match_layout(tmp_path):
    """vec_env.HDR_OFF mirrors struct SssHdr (csrc/sss_layout.h)"""
    from spark_sched_sim_amd.vec_env import HDR_OFF, HDR_PROF

    src = tmp_path / "off.cpp"
    fields = " ".join(f'P({k})' for k in HDR_OFF) + ' P(prof) printf("sizeof %zu\\n", sizeof(SssHdr));'
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "sss_layout.h"\nint main(){\n'
                   '#define P(f) printf("%s %zu\\n", #f, offsetof(SssHdr, f));\n' + fields + "\nreturn 0;}\n")
    exe = tmp_path / "off"
    subprocess.run(["g++", "-I", osp.join(ROOT, "spark_sched_sim_amd", "csrc"), str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout
    got = {l.split()[0]: int(l.split()[1]) for l in out.splitlines()}
    assert got.pop("prof") == HDR_PROF  # the profiling counters VecSparkSchedSimEnv.counters() reads
    assert got.pop("sizeof") == 320
    assert got == HDR_OFF


def test_workload_pack_is_frozen(pack):
    from golden_util import Golden
    from spark_sched_sim_amd import workload

    assert workload.pack_digest(pack) == Golden("c1_fair").pack_sha256
    arrs = workload.build_pack_arrays(workload.make_raw_workload())
    assert arrs["tmpl_stage_off"].size == 155 and int(np.diff(arrs["tmpl_stage_off"]).max()) <= 64


def test_ctypes_structs_match_the_header(tmp_path):
    """every struct of include/sss.h against its ctypes mirror in binding.py: same size and the same
    offset for every field (compiled from the header with gcc, as a C caller would see it)"""
    from spark_sched_sim_amd import binding as B

    pairs = {"sss_cfg": B.SssCfg, "sss_dims": B.SssDims, "sss_buffers": B.SssBuffers, "sss_decima_graph": B.SssDecimaGraph,
             "sss_decima_lists": B.SssDecimaLists, "sss_decima_policy_args": B.SssDecimaPolicyArgs,
             "sss_decima_sample_args": B.SssDecimaSampleArgs, "sss_gnn_args": B.SssGnnArgs, "sss_mlp_args": B.SssMlpArgs, "sss_collect_args": B.SssCollectArgs, "sss_gnn_encode_args": B.SssGnnEncodeArgs, "sss_rows_args": B.SssRowsArgs, "sss_arena_args": B.SssArenaArgs, "sss_arena_array": B.SssArenaArray, "sss_returns_args": B.SssReturnsArgs, "sss_baseline_args": B.SssBaselineArgs, "sss_bit_list_args": B.SssBitListArgs,
             "sss_concat_part": B.SssConcatPart, "sss_concat_args": B.SssConcatArgs, "sss_segcat_args": B.SssSegcatArgs}
    header = open(osp.join(ROOT, "include", "sss.h")).read()
    assert set(re.findall(r"}\s*(sss_[a-z_]+);", header)) == set(pairs), "a struct of the header has no ctypes mirror (or vice versa)"
    lines = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{osp.join(ROOT, "include", "sss.h")}"', "int main(void) {"]
    for cname, cls in pairs.items():
        lines.append(f'  printf("{cname} %zu\\n", sizeof({cname}));')
        for fname, *_ in cls._fields_:
            lines.append(f'  printf("{cname}.{fname} %zu\\n", offsetof({cname}, {fname}));')
    lines += ["  return 0;", "}"]
    src = tmp_path / "abi.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "abi"
    subprocess.run(["gcc", "-o", str(exe), str(src)], check=True)
    got = dict(line.split() for line in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    for cname, cls in pairs.items():
        assert int(got[cname]) == C.sizeof(cls), cname
        for fname, *_ in cls._fields_:
            assert int(got[f"{cname}.{fname}"]) == getattr(cls, fname).offset, f"{cname}.{fname}"
