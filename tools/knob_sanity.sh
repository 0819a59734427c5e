# Off-arms of this round's planner switches still give golden-green models (a knob that rots is worse than no knob).  Run on a GPU box.
cd ${GRAFT_REPO_ROOT:-/root/repo}
for kn in "ENC0_BNFUSE=0" "ENC0_DIRECT=0" "ENC0_WG_SLOTS=1024" ; do
  echo "== $kn"; SEFD_TUNING=$kn timeout 900 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "golden and not fsn" 2>&1 | tail -1
done
for kn in "FSN_WGCAT2=0" "ONES_MFMA=0" "WGRANK=0" "FSN_FB_LANE=1" "FSN_HOLD=1"; do
  echo "== $kn"; SEFD_TUNING=$kn timeout 900 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "fsn or FullSubNet or subband" 2>&1 | tail -1
done
