#!/usr/bin/env python3
"""Registers / scratch / LDS of every kernel of a built library, read from the code objects' metadata (no GPU needed).
usage: tools/kernel_resources.py [efficientspeech_amd/libesmi.so] [--all]     (default: only kernels with scratch, + a summary line)
The .hip_fatbin section holds one clang offload bundle per translation unit; each is unbundled for gfx950 and `llvm-readelf --notes`
prints the AMDGPU metadata (.name, .vgpr_count, .agpr_count, .private_segment_fixed_size, .group_segment_fixed_size)."""
import os, re, subprocess, sys, tempfile
LLVM = "/opt/rocm/lib/llvm/bin"
lib = next((a for a in sys.argv[1:] if not a.startswith("--")), os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "efficientspeech_amd", "libesmi.so"))
show_all = "--all" in sys.argv
with tempfile.TemporaryDirectory() as td:
    fat = os.path.join(td, "fat.bin")
    subprocess.run(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat], check=True)
    d = open(fat, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    pos = [m.start() for m in re.finditer(re.escape(magic), d)] + [len(d)]
    rows = []
    for k in range(len(pos) - 1):
        b, o = os.path.join(td, f"b{k}.bin"), os.path.join(td, f"co{k}.o")
        open(b, "wb").write(d[pos[k]:pos[k + 1]])
        subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={b}", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={o}"],
                       capture_output=True)
        if not os.path.exists(o):
            continue
        notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", o], capture_output=True, text=True).stdout
        for blk in notes.split("  - .agpr_count:")[1:]:
            g = lambda key: (re.search(rf"\.{key}:\s*(\S+)", blk) or [None, "?"])[1]
            rows.append((g("name"), int(re.match(r"\s*(\d+)", blk).group(1)), int(g("vgpr_count")), int(g("private_segment_fixed_size")), int(g("group_segment_fixed_size"))))
names = subprocess.run(["c++filt"], input="\n".join(r[0] for r in rows), capture_output=True, text=True).stdout.split("\n")
rows = sorted({(n.split("(")[0].replace("void ", "").replace("esmi::", ""), a, v, s, l) for n, (_, a, v, s, l) in zip(names, rows)})
print("| kernel | VGPRs (+ AGPRs) | scratch B / lane | static LDS B |\n|---|---:|---:|---:|")
for n, a, v, s, l in rows:
    if show_all or s:
        print(f"| `{n}` | {v}{' + ' + str(a) if a else ''} | {s} | {l} |")
print(f"\n{len(rows)} kernels, {sum(1 for r in rows if r[3])} with scratch")
