set -e
RTHIP_BVH_DEBUG=1 RE_BVH_DEVICE_MIN_TRIS=1 timeout -k 10 100 python3 tools/gpu_bvh_counters.py 3 2>&1 | grep -v "  bin\["
timeout -k 10 300 python -u -m pytest tests/test_device_bvh.py -m gpu -x -q --timeout 120 --timeout-method thread
echo "== sah"; timeout -k 10 200 python3 tools/gpu_bvh_compare.py 4
echo "== ploc"; RTHIP_BVH_BUILDER=ploc timeout -k 10 200 python3 tools/gpu_bvh_compare.py 4
