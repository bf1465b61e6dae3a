"""hipcc -Rpass-analysis=kernel-resource-usage output (stdin) -> one line per kernel."""
import re, sys, subprocess
txt = sys.stdin.read()
cur = None; rows = {}
for line in txt.splitlines():
    m = re.search(r"remark: [^ ]* (?:Function )?Name: (\S+)", line)
    if m:
        cur = m.group(1); rows[cur] = {}; continue
    m = re.search(r"remark:\s+([\w][\w \[\]/]*): (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
names = list(rows)
dem = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.splitlines()
for n, dn in zip(names, dem):
    r = rows[n]
    print(f"{dn.split('(')[0][:48]:48s} VGPR {r.get('VGPRs', -1):4d} AGPR {r.get('AGPRs', -1):4d} SGPR {r.get('TotalSGPRs', -1):4d} "
          f"occ {r.get('Occupancy [waves/SIMD]', -1)} scratch {r.get('ScratchSize [bytes/lane]', -1)} lds {r.get('LDS Size [bytes/block]', -1)}")
