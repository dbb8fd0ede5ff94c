# usage: bash scripts/pmc_w4.sh [lib ...]   - LDS / wait counters of the F(4,3) conv kernel for each library build
export TMPDIR=/tmp; R=$PWD; cd /tmp
LIBS="$@"; [ -z "$LIBS" ] && LIBS=pesr_amd/libpesr_hip.so
for L in $LIBS; do
  tag=$(basename $L .so); O=$R/gpurun_out/w4pmc_$tag; mkdir -p $O
  PESR_HIP_LIB=$R/$L timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/sq -o run -- python3 $R/scripts/profile_w4.py > $O/sq.log 2>&1
  PESR_HIP_LIB=$R/$L timeout 200 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $O/grbm -o run -- python3 $R/scripts/profile_w4.py > $O/grbm.log 2>&1
  python3 $R/scripts/summarize_profiles.py pmc $O/summary.csv $(find $O -name "*counter_collection.csv")
  echo "== $L"; grep "wino4_kernel" $O/summary.csv | grep "LDS\|GRBM\|WAIT\|WAVE_CYCLES"
done
