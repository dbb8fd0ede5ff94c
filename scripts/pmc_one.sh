# usage: bash scripts/pmc_one.sh <tag> <python script> [lib]  - FETCH_SIZE / WRITE_SIZE / GRBM per kernel of one script (separate passes)
export TMPDIR=/tmp; R=$PWD; tag=$1; S=$2; L=${3:-pesr_amd/libpesr_hip.so}; O=$R/gpurun_out/pmc_$tag; mkdir -p $O; cd /tmp
for c in FETCH_SIZE WRITE_SIZE GRBM_GUI_ACTIVE; do
  PESR_HIP_LIB=$R/$L timeout 200 rocprofv3 --pmc $c --output-format csv -d $O/$c -o run -- python3 $R/$S > $O/$c.log 2>&1
done
python3 $R/scripts/summarize_profiles.py pmc $O/summary.csv $(find $O -name "*counter_collection.csv")
cat $O/summary.csv
