"""Per-kernel register / scratch / LDS figures of one .hip file, from hipcc -Rpass-analysis=kernel-resource-usage (development aid):
   python tools/resusage.py lotus_amd/csrc/lvs_rq.hip [-DLVS_TUNING ...] [--grep PATTERN]"""
import re, subprocess, sys
args = sys.argv[1:]
pat = None
if "--grep" in args:
    i = args.index("--grep"); pat = args[i + 1]; del args[i:i + 2]
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Rpass-analysis=kernel-resource-usage",
       "-c", "-o", "/dev/null"] + args
t = subprocess.run(cmd, capture_output=True, text=True).stderr
for b in t.split("Function Name: ")[1:]:
    name = b.split()[0]
    name = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    if pat and not re.search(pat, name):
        continue
    f = lambda k: re.search(k + r": (\d+)", b).group(1)
    print(f"V {f('VGPRs'):>3} A {f('AGPRs'):>3} S {f('TotalSGPRs'):>3} scratch {f('ScratchSize .bytes/lane.'):>4} lds {f('LDS Size .bytes/block.'):>6} occ {f('Occupancy .waves/SIMD.')}  {name[:150]}")
if "error" in t:
    print("\n".join(l for l in t.splitlines() if "error" in l)[:2000])
