#!/bin/bash
# frames in flight: auxiliary kernels at raised wave priority (two builds) x compositor priority x compositor pool
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
L=$PWD/splatapult_amd/lib
run() {
  tag=$1; shift
  env "$@" timeout 300 python bench.py --steps 500 --warmup 50 --no-cpu-baseline --serial-frames 8 --profile-frames 1 > gpurun_out/r2i_$tag.json 2> gpurun_out/r2i_$tag.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/r2i_$tag.json").read().strip().splitlines()[-1])
    print("%-22s fps %7.1f ms %.4f | overlapped compk %.4f" % ("$tag", d["value"], d["ms_per_step"], d["stages_ms"].get("composite_kernel", 0)))
except Exception as e:
    print("$tag failed:", e); print(open("gpurun_out/r2i_$tag.err").read()[-800:])
PY
}
run base X=1
run base_p0 MSPLAT_COMP_PRIO=0
run aux3_p1 MSPLAT_LIB_PATH=$L/ab_auxprio3.so
run aux3_p0 MSPLAT_LIB_PATH=$L/ab_auxprio3.so MSPLAT_COMP_PRIO=0
run aux3_p0_w2048 MSPLAT_LIB_PATH=$L/ab_auxprio3.so MSPLAT_COMP_PRIO=0 MSPLAT_COMP_WAVES=2048
run aux3_p0_w4096 MSPLAT_LIB_PATH=$L/ab_auxprio3.so MSPLAT_COMP_PRIO=0 MSPLAT_COMP_WAVES=4096
run aux3_p0_w8192 MSPLAT_LIB_PATH=$L/ab_auxprio3.so MSPLAT_COMP_PRIO=0 MSPLAT_COMP_WAVES=8192
run aux1_p0 MSPLAT_LIB_PATH=$L/ab_auxprio1.so MSPLAT_COMP_PRIO=0
run aux1_p0_w2048 MSPLAT_LIB_PATH=$L/ab_auxprio1.so MSPLAT_COMP_PRIO=0 MSPLAT_COMP_WAVES=2048
run base2 X=1
