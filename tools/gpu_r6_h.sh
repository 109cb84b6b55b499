#!/bin/bash
# (the ablation code these variants were built from is in commit 8eef668; tools/build_variant.sh <name> -D...)
# round 6, run h: what does each stage COST with four frames in flight?  Ablation builds (wrong pixels on purpose, same structure):
# half the projection's record bytes, every second record skipped in the compositor's inner loop, the sort stopped after pass 0,
# no compositor at all -- frames/s in flight and ms serial against the product build, same box
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out
fps() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); s=d['serial']; st=s['stages_ms']; f=d['stages_ms']
        print('%-22s in flight %.4f ms/frame (%.0f fps)  serial %.4f ms | serial us: sort %.1f project %.1f binning %.1f compk %.1f | in flight us: sort %.0f project %.0f binning %.0f compk %.0f' % ('$1', d['ms_per_step'], d['value'], s['ms_per_frame'], 1e3*st['sort_total'], 1e3*st['project'], 1e3*st['binning'], 1e3*st['composite_kernel'], 1e3*f['sort_total'], 1e3*f['project'], 1e3*f['binning'], 1e3*f['composite_kernel']))
"; }
V=$PWD/tools/bin/variants
run() {  # label, env... -- (bench args)
  label=$1; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --steps 200 --warmup 50 --serial-frames 64 --two-pass off 2>>gpurun_out/r06h_err.txt | fps "$label"
}
for rep in 1 2; do
  run product X=1
  run proj_half_bytes MSPLAT_LIB_PATH=$V/libmsplat_projhalf.so
  run comp_half_evals MSPLAT_LIB_PATH=$V/libmsplat_comphalf.so
  run sort_pass0_only MSPLAT_LIB_PATH=$V/libmsplat_ablate.so MSPLAT_X_SORT_SKIP=1
  run no_compositor MSPLAT_LIB_PATH=$V/libmsplat_ablate.so MSPLAT_X_COMP_NONE=1
  run no_comp_sort0 MSPLAT_LIB_PATH=$V/libmsplat_ablate.so MSPLAT_X_COMP_NONE=1 MSPLAT_X_SORT_SKIP=1
done
tail -3 gpurun_out/r06h_err.txt
