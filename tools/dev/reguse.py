#!/usr/bin/env python3
"""Register / LDS / spill table of every kernel of one .hip file (compile only: no GPU needed).
usage: tools/dev/reguse.py flappie_amd/csrc/ffhip_rnn_split.hip [filter] [-- extra hipcc flags]"""
import re, subprocess, sys
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 and sys.argv[2] != "--" else ""
extra = sys.argv[sys.argv.index("--") + 1:] if "--" in sys.argv else []
noslp = [] if ("rnn_split" in src or "_exp" in src) else ["-fno-slp-vectorize"]
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-I", "flappie_amd/csrc", "-c", src,
       "-o", "/tmp/reguse.o", "-Rpass-analysis=kernel-resource-usage"] + noslp + extra
err = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], None
for line in err.splitlines():
    m = re.search(r"remark: +(Function Name|Name): (\S+)", line)
    if m:
        cur = {"name": subprocess.run(["c++filt", m.group(2)], capture_output=True, text=True).stdout.strip()}
        rows.append(cur)
        continue
    m = re.search(r"remark: +([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
    if m and cur is not None:
        cur[m.group(1).strip()] = int(m.group(2))
    elif "error" in line:
        print(line)
print(f"{'kernel':70s} VGPR AGPR spill scratch occ   LDS")
for r in rows:
    if flt in r["name"]:
        n = re.sub(r"\(.*", "", r["name"]).replace("ffhip::", "")
        print(f"{n:70s} {r.get('VGPRs', -1):4d} {r.get('AGPRs', -1):4d} {r.get('VGPRs Spill', -1):5d} {r.get('ScratchSize', -1):7d} {r.get('Occupancy', -1):3d} {r.get('LDS Size', -1):6d}")
