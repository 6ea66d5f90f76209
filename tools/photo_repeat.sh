#!/bin/bash
# the sample photograph's --model_name none inversion N times, each in a process of its own: the loop-1 'rm' line of its log and the final PSNR
# (a hunt for run-to-run differences)   usage: tools/photo_repeat.sh [N]
cd "$GRAFT_REPO_ROOT" || exit 1
for i in $(seq 1 ${1:-12}); do
  rm -rf /tmp/pr$i; timeout 120 python tools/real_image.py --sample indoor2 --model_name none --out /tmp/pr$i > /dev/null 2>&1
  python - <<PY
import json, glob
d = json.load(open(glob.glob("/tmp/pr$i/real_image_*.json")[0]))
print($i, d["psnr_vs_photo"]["this_build_final_render"], d["log"][2].split("] ")[1])
PY
done
