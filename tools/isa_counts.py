"""Instruction-class counts of the simulator kernels' gfx950 code ("dump the ISA with
VALU / SALU / LDS counts"). Compiles csrc/sss_hip.hip with -save-temps into a scratch directory and parses
the device assembly: per kernel the static instruction mix, registers, LDS and scratch.

usage: python tools/isa_counts.py [--out profiles/r02_isa.md] [--unit sss_hip_sim.hip | sss_hip_wide.hip]

`kernel_metadata(so)` reads registers / LDS / scratch / spill counts of every kernel straight from a BUILT library's code objects
(seconds, no recompile): tests/test_abi.py holds the simulator kernels' scratch sizes with it."""
import argparse, collections, os, os.path as osp, re, subprocess, sys, tempfile

ROOT = osp.dirname(osp.dirname(osp.abspath(__file__)))
CLASSES = [("VALU", r"^v_(?!readlane|readfirstlane|writelane)"), ("cross-lane (readlane/writelane/DPP moves)", r"^v_(readlane|readfirstlane|writelane)"),
           ("SALU", r"^s_(?!waitcnt|load|buffer_load|barrier|branch|cbranch|nop|endpgm|sleep|setprio|memtime|memrealtime)"),
           ("scalar memory", r"^s_(load|buffer_load|memtime|memrealtime)"), ("LDS", r"^ds_"), ("global / flat memory", r"^(global|flat|buffer|scratch)_"),
           ("s_waitcnt", r"^s_waitcnt"), ("branches", r"^s_(branch|cbranch)"), ("other", r".")]


LLVM_BIN = "/opt/rocm/lib/llvm/bin"


def compile_asm(tmp, unit="sss_hip_sim.hip"):
    """the unit's device assembly, compiled with the product build's own flags (spark_sched_sim_amd/build.py)"""
    sys.path.insert(0, ROOT)
    from spark_sched_sim_amd import build
    src = osp.join(ROOT, "spark_sched_sim_amd", "csrc", unit)
    cmd = [build.hipcc()] + [f for f in build.FLAGS if f != "-Wall"] + build.UNIT_FLAGS.get(unit, []) + ["-c", "-save-temps", "-o", osp.join(tmp, "unit.o"), src]
    subprocess.run(cmd, cwd=tmp, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return osp.join(tmp, osp.splitext(unit)[0] + "-hip-amdgcn-amd-amdhsa-gfx950.s")


def kernel_metadata(so: str) -> dict:
    """kernel name -> {private_segment_fixed_size (scratch bytes per lane), vgpr_count, sgpr_count, group_segment_fixed_size (static LDS),
    vgpr_spill_count, sgpr_spill_count} for every gfx950 kernel of a built library or object file: the .hip_fatbin section's offload
    bundles (one per translation unit) are unbundled and their AMDGPU metadata notes parsed"""
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        fb = osp.join(tmp, "fatbin")
        subprocess.run([osp.join(LLVM_BIN, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", so, fb], check=True)
        data = open(fb, "rb").read()
        starts = [m.start() for m in re.finditer(re.escape(b"__CLANG_OFFLOAD_BUNDLE__"), data)]
        for i, st in enumerate(starts):
            part, co = osp.join(tmp, f"bundle{i}"), osp.join(tmp, f"code{i}.co")
            open(part, "wb").write(data[st:(starts[i + 1] if i + 1 < len(starts) else len(data))])
            subprocess.run([osp.join(LLVM_BIN, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + part,
                            "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], check=True, stderr=subprocess.DEVNULL)
            notes = subprocess.run([osp.join(LLVM_BIN, "llvm-readelf"), "--notes", co], capture_output=True, text=True, check=True).stdout
            for blk in re.split(r"\n\s*- \.agpr_count", notes)[1:]:  # one block per kernel
                d = {}
                for k in ("name", "private_segment_fixed_size", "vgpr_count", "sgpr_count", "group_segment_fixed_size", "vgpr_spill_count", "sgpr_spill_count"):
                    m = re.search(r"\." + k + r":\s+(\S+)", blk)
                    if m:
                        d[k] = m.group(1)
                out[d["name"]] = {k: int(v) for k, v in d.items() if k != "name"}
    return out


def kernels(asm):
    """name -> list of instructions; name -> {.amdhsa_ directive: value}"""
    out, cur, meta, kd = {}, None, {}, None
    for line in open(asm):
        t = line.strip()
        m = re.match(r"^\.amdhsa_kernel\s+(\w+)", t)
        if m:
            kd = m.group(1)
            meta[kd] = {}
            continue
        if t.startswith(".end_amdhsa_kernel"):
            kd = None
            continue
        if kd:
            m = re.match(r"^\.amdhsa_(next_free_vgpr|next_free_sgpr|group_segment_fixed_size|private_segment_fixed_size|accum_offset)\s+(\S+)", t)
            if m:
                meta[kd][m.group(1)] = m.group(2)
            continue
        m = re.match(r"^(\w+):\s*(;.*)?$", t)
        if m and not m.group(1).startswith(("BB", "LBB")):
            cur = m.group(1)
            out[cur] = []
            continue
        if t.startswith(".Lfunc_end"):
            cur = None
            continue
        if cur and t and not t.startswith((";", ".", "//")) and not t.endswith(":"):
            out[cur].append(t.split(";")[0].strip())
    return out, meta


def mix(instrs):
    c = collections.Counter()
    for i in instrs:
        op = i.split()[0]
        for name, pat in CLASSES:
            if re.match(pat, op):
                c[name] += 1
                break
    c["DPP-modified VALU"] = sum(1 for i in instrs if "row_" in i or "quad_perm" in i or "wave_" in i)
    return c


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--unit", default="sss_hip_sim.hip")
    a = ap.parse_args()
    with tempfile.TemporaryDirectory() as tmp:
        asm = compile_asm(tmp, a.unit)
        ks, meta = kernels(asm)
    lines = ["# gfx950 instruction mix of the simulator kernels (static counts, `tools/isa_counts.py`)", "",
             "`hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -save-temps`; every procedure is inlined into the kernels (the", "launch context is read through the kernel-argument segment pointer, which callees do not have).", ""]
    names = [k for k in ks if k.startswith("sss_") and (k.endswith("_kernel") or k.endswith("_kernel_wide"))]
    cols = [n for n, _ in CLASSES] + ["DPP-modified VALU"]
    lines.append("| kernel | instructions | " + " | ".join(cols) + " | VGPRs | SGPRs | static LDS (B) | scratch (B) |")
    lines.append("|---|---|" + "---|" * (len(cols) + 4))
    for k in names:
        c = mix(ks[k])
        md = meta.get(k, {})
        lines.append(f"| `{k}` | {len(ks[k])} | " + " | ".join(str(c[n]) for n in cols) +
                     f" | {md.get('next_free_vgpr', '?')} | {md.get('next_free_sgpr', '?')} | {md.get('group_segment_fixed_size', '?')} | {md.get('private_segment_fixed_size', '?')} |")
    pmc = osp.join(ROOT, "profiles", "r03_pmc_step_c2.txt")
    if osp.exists(pmc):
        lines += ["", "Dynamic counts of `sss_step_kernel` (rocprofv3 PMC passes over a C2 step-mode run, 4096 waves per launch, mean of the last 40",
                  "launches; `tools/debug/pmc_probe.sh`):", "", "```"] + open(pmc).read().rstrip().split("\n") + ["```"]
    text = "\n".join(lines) + "\n"
    if a.out:
        open(osp.join(ROOT, a.out) if not osp.isabs(a.out) else a.out, "w").write(text)
    print(text)


if __name__ == "__main__":
    main()
