for l in "$@"; do echo "== $l"; PESR_HIP_LIB=$l python scripts/rgb_layer_time.py 2>&1 | grep "dedicated"; done
