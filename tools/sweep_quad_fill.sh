set -e
for f in 0 2048 5120 8192 16384; do
  echo "== RTHIP_QUAD_FILL=$f"
  RTHIP_QUAD_FILL=$f timeout -k 10 120 python tools/gpu_floor.py 3 2>&1 | grep "replayed\|1920x1080 plain\|   level"  | head -12
  RTHIP_QUAD_FILL=$f RANKSHARE_F=2 RANKSHARE_B=10 RANKSHARE_K=20 RANKSHARE_ROTATE=1 RANKSHARE_REPEAT=21 timeout -k 10 120 python tools/gpu_rank_share.py 8 2>&1 | grep parts
done
