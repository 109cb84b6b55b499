#!/bin/bash
# SQ / GRBM counters of EVERY kernel of the serial frame (config 2): how busy the vector units are in the latency-bound chain.
# One rocprofv3 --pmc pass per counter group, kernel trace only (no other trace domains).
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
OUT=gpurun_out/r03_pmc_sq_cfg2.txt
echo "# rocprofv3 --pmc SQ / GRBM counters per dispatch, bench.py --frames-in-flight 1, config 2 (tools/gpu_round3_u2.sh; one counter group per pass)" > $OUT
echo "# SQ_* cycle counters are in quad-cycles (x4 = cycles) and summed over the chip; GRBM_GUI_ACTIVE is summed over the 8 XCDs" >> $OUT
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  bash tools/gpu_pmc.sh r3u_$i "$grp" --frames-in-flight 1 --prewarm 20 --serial-frames 8 2>&1 | grep -E "^kernel |msplat::" | cut -c1-260 >> $OUT
done
python - <<'PY'
import re
rows = {}
hdr = None
for line in open("gpurun_out/r03_pmc_sq_cfg2.txt"):
    if line.startswith("#"): continue
    parts = line.split()
    if line.startswith("kernel"):
        hdr = [p for p in parts[1:] if p.isupper() or "_" in p]
        hdr = [h for h in hdr if h not in ("(per", "dispatch)")]
        continue
    name = line[:50].strip()
    vals = line[50:].split()
    d = rows.setdefault(name, {})
    for h, v in zip(hdr, vals):
        try: d[h] = float(v)
        except ValueError: pass
out = open("gpurun_out/r03_pmc_sq_cfg2.txt", "a")
out.write("# derived per dispatch: duration = GRBM_GUI_ACTIVE / 8 cycles; VALU-busy = 4 x SQ_ACTIVE_INST_VALU / (1024 SIMDs x duration);\n")
out.write("# resident waves per SIMD (time average) = 4 x SQ_WAVE_CYCLES / (1024 x duration); cycles per VALU instruction = 4 x SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU\n")
out.write("%-50s %10s %10s %12s %12s\n" % ("kernel", "cycles", "VALU-busy", "waves/SIMD", "clk/VALU"))
for k, d in sorted(rows.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0)):
    if "GRBM_GUI_ACTIVE" not in d or "SQ_ACTIVE_INST_VALU" not in d: continue
    dur = d["GRBM_GUI_ACTIVE"] / 8.0
    line = "%-50s %10.0f %9.1f%% %12.2f %12.2f" % (k, dur, 100 * 4 * d["SQ_ACTIVE_INST_VALU"] / (1024 * dur), 4 * d.get("SQ_WAVE_CYCLES", 0) / (1024 * dur), 4 * d["SQ_ACTIVE_INST_VALU"] / max(d.get("SQ_INSTS_VALU", 1), 1))
    out.write(line + "\n"); print(line)
PY
