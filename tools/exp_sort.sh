for v in 0 1 4 5 3 7; do echo "== RTHIP_SORT_HARD=$v"; RTHIP_SORT_HARD=$v timeout -k 10 120 python3 tools/gpu_levels.py 4 || exit 1; done
