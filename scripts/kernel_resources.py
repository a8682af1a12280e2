"""Static resource usage of the shipped kernels: `hipcc -Rpass-analysis=kernel-resource-usage` over the library's translation units with the
flags of fadtk_amd/build.py, one line per kernel (VGPRs, AGPRs, scratch bytes per lane, static LDS bytes, waves per SIMD).
    python scripts/kernel_resources.py [substring ...] > profiles/rNN_kernel_resources.txt
No GPU needed (hipcc cross-compiles gfx950)."""
import re, subprocess, sys, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
from fadtk_amd import build as B

want = sys.argv[1:]
src = sorted((pathlib.Path(B.__file__).parent / "csrc").glob("*.hip"))
print("# hipcc -Rpass-analysis=kernel-resource-usage (gfx950, the flags of fadtk_amd/build.py): registers, scratch, LDS and occupancy")
print("kernel | VGPRs | AGPRs | scratch B/lane | LDS B (static) | waves/SIMD | file")
for s in src:
    r = subprocess.run([B._hipcc(), *B.FLAGS, "-Rpass-analysis=kernel-resource-usage", "-c", str(s), "-o", "/dev/null"], capture_output=True, text=True)
    cur = {}
    for line in r.stderr.splitlines():
        m = re.search(r"remark:\s+(Function Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|LDS Size \[bytes/block\]|Occupancy \[waves/SIMD\]): (\S+)", line)
        if not m: continue
        k, v = m.group(1), m.group(2).strip()
        if k == "Function Name":
            cur = {"name": v}
        else:
            cur[k.split(" ")[0]] = v
        if k.startswith("LDS"):
            name = subprocess.run(["c++filt", cur["name"]], capture_output=True, text=True).stdout.strip()
            name = re.sub(r"\(.*\)$", "", name)
            if "anchor" in name or (want and not any(w in name for w in want)): continue
            print(f"{name} | {cur.get('VGPRs')} | {cur.get('AGPRs')} | {cur.get('ScratchSize')} | {cur.get('LDS')} | {cur.get('Occupancy')} | {s.name}")
