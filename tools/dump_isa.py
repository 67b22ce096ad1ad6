#!/usr/bin/env python3
"""Kernel resource table + hot-loop disassembly straight from the built objects (no GPU needed).

  python tools/dump_isa.py OUTDIR [regex ...]

For every avif-format_amd/build/*.hip.o: pulls the gfx950 code object out of .hip_fatbin (objcopy + clang-offload-bundler),
reads the AMDGPU metadata notes (VGPRs, AGPRs, SGPRs, LDS, scratch, occupancy derived from VGPRs and -- since round 6 -- SGPRs), and writes
  OUTDIR/resources.tsv   one row per kernel (demangled name, vgpr, agpr, sgpr, lds bytes, scratch bytes, waves/SIMD by VGPR and SGPR)
  OUTDIR/<n>.s           llvm-objdump -d of every kernel whose demangled name matches one of the regexes
so that statements like "54 VGPRs, 8 waves/SIMD, 6 x global_load_dwordx4 nt" in DESIGN.md can be checked against files."""
import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def sh(*a, **k):
    return subprocess.run(a, check=True, stdout=subprocess.PIPE, text=True, **k).stdout


def main():
    out = sys.argv[1]
    pats = [re.compile(p) for p in sys.argv[2:]]
    os.makedirs(out, exist_ok=True)
    rows = []
    for obj in sorted(glob.glob(os.path.join(os.environ.get("ISA_OBJDIR", os.path.join(ROOT, "avif-format_amd", "build")), "*.hip.o"))):
        base = os.path.basename(obj).split(".")[0]
        fat, co = f"/tmp/{base}.fat", f"/tmp/{base}.co"
        subprocess.run(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, fat], check=True)
        if not os.path.getsize(fat):
            continue
        subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}",
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], check=True)
        notes = sh(f"{LLVM}/llvm-readelf", "--notes", co)
        kernels, cur = [], None
        for line in notes.splitlines():
            m = re.match(r"\s+(- )?\.(\w+):\s*(.*)", line)
            if not m:
                continue
            if m.group(1) and m.group(2) == "agpr_count":        # first key of a kernel record
                cur = {}
                kernels.append(cur)
            if cur is not None and m.group(2) not in cur:
                cur[m.group(2)] = m.group(3).strip().strip("'")
        kernels = [k for k in kernels if "name" in k and "vgpr_count" in k]
        names = subprocess.run(["c++filt"], input="\n".join(k["name"] for k in kernels), stdout=subprocess.PIPE, text=True,
                               check=True).stdout.splitlines() if kernels else []
        dis = None
        for k, dn in zip(kernels, names):
            vg, ag = int(k.get("vgpr_count", 0)), int(k.get("agpr_count", 0))
            tot = max(vg + ag, 1)                      # unified 512-entry file per SIMD lane, allocation granule 8
            sg = int(k.get("sgpr_count", 0))
            # ... and the scalar file (round 6): MI355X admits min(8, 800 // (ceil(sgpr / 16) * 16 + 16)) waves per SIMD
            # (MI355X_MICROARCH.md, "Residency": <= 80 SGPRs -> 8, 82-96 -> 7, 98-112 -> 6) -- rounds 1-5 looked at VGPRs only
            waves = min(8, 512 // (((tot + 7) // 8) * 8), 800 // (((sg + 15) // 16) * 16 + 16) if sg else 8)
            rows.append((base, dn, vg, ag, int(k.get("sgpr_count", 0)), int(k.get("group_segment_fixed_size", 0)),
                         int(k.get("private_segment_fixed_size", 0)), waves))
            if any(p.search(dn) for p in pats):
                if dis is None:
                    dis = sh(f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", co)
                m = re.search(r"^[0-9a-f]+ <" + re.escape(k["name"]) + r">:\n(.*?)(?=^[0-9a-f]+ <|\Z)", dis, re.S | re.M)
                if m:
                    fn = re.sub(r"[^A-Za-z0-9_.=,-]+", "_", dn)[:150]
                    body = re.sub(r"[ \t]*//[ ]*[0-9A-Fa-f]{8,}:.*$", "", m.group(1), flags=re.M)     # drop address / encoding comments
                    ops = re.findall(r"^\s+(\w+)", body, re.M)
                    hist = {}
                    for o in ops:
                        hist[o] = hist.get(o, 0) + 1
                    top = sorted(hist.items(), key=lambda kv: -kv[1])[:25]
                    with open(os.path.join(out, fn + ".s"), "w") as f:
                        f.write(f"; {dn}\n; vgpr {vg} agpr {ag} sgpr {k.get('sgpr_count')} lds {k.get('group_segment_fixed_size')} B "
                                f"scratch {k.get('private_segment_fixed_size')} B -> {waves} waves/SIMD by VGPRs\n"
                                f"; {len(ops)} instructions; most frequent: " + ", ".join(f"{o} x{n}" for o, n in top) + "\n")
                        f.write(body)
    with open(os.path.join(out, "resources.tsv"), "w") as f:
        f.write("unit\tkernel\tvgpr\tagpr\tsgpr\tlds_bytes\tscratch_bytes\twaves_per_simd_by_vgpr\n")
        for r in rows:
            f.write("\t".join(str(x) for x in r) + "\n")
    print(f"{len(rows)} kernels -> {out}/resources.tsv")


if __name__ == "__main__":
    main()
