#!/bin/bash
# round 6, run a: where the four-frames-in-flight gap goes, measured instead of guessed.
#  1. driver-protocol baseline of this tree on this box (x2)
#  2. SQ / TCC counters per kernel with four frames in flight AND serial (rocprofv3 --pmc, one group per pass; the dispatch durations
#     in the counter records tell whether rocprofv3 serialised the dispatches)
#  3. PC sampling (stochastic, then host-trap) with four frames in flight: does not serialise
#  4. the compositor's own probe, serial against in flight (tools/inflight_probe.py)
#  5. what a chain shared by two views buys in flight at config 2's size (cfg2v2, stereo batch on / off)
cd ${GRAFT_REPO_ROOT:-.}
R=$(pwd)
export TMPDIR=/tmp
mkdir -p gpurun_out
T=r06a
fps() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); s=d.get('serial',{})
        print('$1 fps %.0f  ms/step %.4f  serial %.4f ms  in flight us: sort %.0f project %.0f binning %.0f composite_kernel %.0f' % (d['value'], d['ms_per_step'], s.get('ms_per_frame',0), 1e3*d['stages_ms']['sort_total'], 1e3*d['stages_ms']['project'], 1e3*d['stages_ms']['binning'], 1e3*d['stages_ms']['composite_kernel']))
"; }
for rep in 1 2; do timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>gpurun_out/${T}_err.txt | fps base_steps20; done
timeout 300 python bench.py --no-cpu-baseline --steps 200 --warmup 50 2>>gpurun_out/${T}_err.txt | fps base_steps200

pmc() {  # tag, counters, bench args...
  tag=$1; ctrs=$2; shift; shift
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $ctrs -d $R/gpurun_out/${T}_pmc_$tag -o run --output-format csv -- \
     python $R/bench.py --no-cpu-baseline --profile-frames 1 --timing-stride 0 --prewarm 40 "$@" > $R/gpurun_out/${T}_pmc_$tag.log 2>&1)
  grep -h "^{" gpurun_out/${T}_pmc_$tag.log | fps "under_pmc_$tag"
}
G1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY"
G2="TCC_HIT_sum TCC_MISS_sum TCC_EA_RDREQ_sum TCC_EA_WRREQ_sum"
G3="SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_INSTS_LDS"
for g in 1 2 3; do
  eval "C=\$G$g"
  pmc fif4_g$g "$C" --steps 60 --warmup 20 --serial-frames 8
  pmc serial_g$g "$C" --frames-in-flight 1 --steps 60 --warmup 20 --serial-frames 8
done
for m in fif4 serial; do
  python tools/pmc_table.py "$m: rocprofv3 --pmc over bench.py ($m), config 2" $(find gpurun_out/${T}_pmc_${m}_g* -name run_counter_collection.csv) > gpurun_out/${T}_pmc_sq_$m.txt 2>&1
  head -60 gpurun_out/${T}_pmc_sq_$m.txt | cut -c1-200
done
rm -rf gpurun_out/${T}_pmc_*_g*/

# 3. PC sampling
for method in stochastic host_trap; do
  unit=cycles; iv=1048576
  [ $method = host_trap ] && { unit=time; iv=100; }
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pc-sampling-beta-enabled --pc-sampling-method $method --pc-sampling-unit $unit --pc-sampling-interval $iv \
     -d $R/gpurun_out/${T}_pcs_$method -o run --output-format csv -- \
     python $R/bench.py --no-cpu-baseline --profile-frames 1 --timing-stride 0 --prewarm 40 --steps 100 --warmup 20 --serial-frames 8 > $R/gpurun_out/${T}_pcs_$method.log 2>&1)
  echo "pc sampling $method: rc $?"; tail -3 gpurun_out/${T}_pcs_$method.log | cut -c1-300
  grep -h "^{" gpurun_out/${T}_pcs_$method.log | fps "under_pcs_$method"
  ls -la $(find gpurun_out/${T}_pcs_$method -type f | head -8) 2>/dev/null
  python tools/pcsamp_summary.py gpurun_out/${T}_pcs_$method > gpurun_out/${T}_pcsamp_$method.txt 2>&1
  head -50 gpurun_out/${T}_pcsamp_$method.txt | cut -c1-260
  rm -rf gpurun_out/${T}_pcs_$method
done

# 4. the compositor's probe, serial against in flight
timeout 300 python tools/inflight_probe.py 4 240 > gpurun_out/${T}_inflight_probe.txt 2>&1; cat gpurun_out/${T}_inflight_probe.txt | tail -8

# 5. two views of config 2's size: one chain for both against one chain per view
for rep in 1 2; do
  timeout 300 python bench.py --no-cpu-baseline --workload cfg2v2 --steps 20 --warmup 5 2>>gpurun_out/${T}_err.txt | fps cfg2v2_one_chain
  timeout 300 python bench.py --no-cpu-baseline --workload cfg2v2 --steps 20 --warmup 5 --no-stereo-batch 2>>gpurun_out/${T}_err.txt | fps cfg2v2_two_chains
done
tail -5 gpurun_out/${T}_err.txt
