"""where a kernel's scratch (spill) instructions come from: python scripts/debug/spill_lines.py <tu>.hip <mangled-name-prefix>
(compiles the unit with -gline-tables-only -S and attributes scratch_* instructions to source lines)"""
import re, subprocess, sys, os
tu, pref = sys.argv[1], sys.argv[2]
out = "/tmp/spill_" + os.path.basename(tu) + ".s"
if not (len(sys.argv) > 3 and sys.argv[3] == "reuse"):
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-gline-tables-only", "-S", "--cuda-device-only", tu, "-o", out],
                   check=True, stderr=subprocess.DEVNULL)
lines = open(out).read().split("\n")
start = [i for i, l in enumerate(lines) if l.startswith(pref)][0]
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
cur, cnt, tot = None, {}, {}
for l in lines[start:end]:
    m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)\s+(\d+)", l)
    if m:
        cur = (int(m.group(1)), int(m.group(2)))
        continue
    t = l.strip()
    if not t or t.startswith(".") or t.startswith(";"):
        continue
    tot[cur] = tot.get(cur, 0) + 1
    if "scratch_" in l:
        cnt[cur] = cnt.get(cur, 0) + 1
print("instructions", sum(tot.values()), "scratch", sum(cnt.values()))
print("scratch by (file, line):", sorted(cnt.items(), key=lambda x: -x[1])[:24])
print("instructions by (file, line):", sorted(tot.items(), key=lambda x: -x[1])[:16])
