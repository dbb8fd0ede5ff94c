#!/bin/bash
# ONE parametrised wrapper for the gpurun batches of a round (replaces the per-call scripts/gpu_call<N>.sh of rounds 1-3).
# Run from the repo root on the GPU box:   gpurun --timeout S -- 'bash scripts/gpu_job.sh <job> [args] ; bash scripts/gpu_job.sh <job> ...'
#   tests [pytest args]         the -m gpu suite (log -> gpurun_out/<tag>/tests.log)
#   bench <name> [bench args]   one bench.py line -> gpurun_out/<tag>/<name>.json
#   dp1 <name> [bench args]     the same under a ONE-rank RCCL group with PESR_FORCE_DP=1 (every hook / bucket / RCCL call)
#   stats [bench args]          rocprofv3 --kernel-trace --stats of a short bench run -> kernel_stats.csv
#   trace                       single-stream kernel trace condensed per (kernel, grid) -> kernel_trace_by_grid.csv
#   pmc <name> <script.py>      SQ / GRBM / FETCH_SIZE / WRITE_SIZE in four SEPARATE passes over one script -> <name>_pmc_summary.csv
#   hbm <name> <script.py>      FETCH_SIZE / WRITE_SIZE / GRBM_GUI_ACTIVE only (HBM-bound kernels)
#   tcc <name> <script.py>      TCC_EA0_RDREQ / _32B / TCC_HIT / TCC_MISS (+ TCC_REQ / READ / WRITE in a second pass) -> <name>_tcc_summary.csv
#   py <name> <script.py> [..]  any measurement script, output -> <name>.txt
# TAG=<dir> selects the output directory under gpurun_out/ (default: job).
export TMPDIR=/tmp; export HSA_ENABLE_IPC_MODE_LEGACY=0
R=$PWD; O=$R/gpurun_out/${TAG:-job}; mkdir -p $O
job=$1; shift
case $job in
  tests)  cd $R; timeout 1500 python3 -m pytest tests -m gpu -x -q "$@" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -5 $O/tests.log ;;
  bench)  n=$1; shift; cd /tmp; timeout 900 python3 $R/bench.py "$@" 2> $O/$n.err | tail -1 > $O/$n.json; cut -c1-400 $O/$n.json; echo; tail -3 $O/$n.err ;;
  dp1)    n=$1; shift; cd /tmp
          PESR_FORCE_DP=1 timeout 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29521 \
              $R/bench.py --gpus 1 "$@" 2> $O/$n.err | tail -1 > $O/$n.json; cut -c1-400 $O/$n.json; echo; tail -3 $O/$n.err ;;
  stats)  cd /tmp; timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o run -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-side "$@" > $O/stats.log 2>&1
          cp $(find $O/stats -name "*kernel_stats.csv") $O/kernel_stats.csv 2>/dev/null; rm -rf $O/stats; head -25 $O/kernel_stats.csv ;;
  trace)  cd /tmp; timeout 400 rocprofv3 --kernel-trace --output-format csv -d $O/single -o run -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-events --no-side > $O/single.log 2>&1
          cd $R; python3 scripts/summarize_profiles.py trace $(find $O/single -name "*kernel_trace.csv") 6 $O/kernel_trace_by_grid.csv 2
          rm -rf $O/single; head -40 $O/kernel_trace_by_grid.csv ;;
  pmc)    n=$1; s=$2; cd /tmp
          timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/pmc_sq -o run -- python3 $R/$s > $O/pmc_sq.log 2>&1
          for c in GRBM_GUI_ACTIVE FETCH_SIZE WRITE_SIZE; do timeout 200 rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -o run -- python3 $R/$s > $O/pmc_$c.log 2>&1; done
          cd $R; python3 scripts/summarize_profiles.py pmc $O/${n}_pmc_summary.csv $(find $O/pmc_* -name "*counter_collection.csv"); rm -rf $O/pmc_*/
          grep -v "at::native" $O/${n}_pmc_summary.csv | head -80 ;;
  hbm)    n=$1; s=$2; cd /tmp
          for c in FETCH_SIZE WRITE_SIZE GRBM_GUI_ACTIVE; do timeout 300 rocprofv3 --pmc $c --output-format csv -d $O/hbm_$c -o run -- python3 $R/$s > $O/hbm_$c.log 2>&1; done
          cd $R; python3 scripts/summarize_profiles.py pmc $O/${n}_pmc_summary.csv $(find $O/hbm_* -name "*counter_collection.csv"); rm -rf $O/hbm_*/
          grep -v "at::native" $O/${n}_pmc_summary.csv | head -80 ;;
  tcc)    n=$1; s=$2; cd /tmp      # L2 <-> fabric request counters (one pass: four TCC slots): read requests, the 32-byte ones among them, L2 hits / misses
          timeout 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/tcc_a -o run -- python3 $R/$s > $O/tcc_a.log 2>&1
          timeout 300 rocprofv3 --pmc TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum --output-format csv -d $O/tcc_b -o run -- python3 $R/$s > $O/tcc_b.log 2>&1
          cd $R; python3 scripts/summarize_profiles.py pmc $O/${n}_tcc_summary.csv $(find $O/tcc_* -name "*counter_collection.csv"); rm -rf $O/tcc_*/
          grep -v "at::native" $O/${n}_tcc_summary.csv | head -40; tail -3 $O/tcc_a.log ;;
  py)     n=$1; s=$2; shift 2; cd $R; timeout 900 python3 $s "$@" 2>&1 | grep -v amdgpu.ids > $O/$n.txt; tail -60 $O/$n.txt ;;
  *)      echo "unknown job $job"; exit 2 ;;
esac
