# A/B of wino_gemm_x3w_kernel forms per layer shape (A3D_X3W_WM = 2: 64-tile blocks, 4: 128-tile blocks) and, with a library built
# with A3D_HIPCC_FLAGS=-DA3D_ABLATIONS, timing-only ablations of the 128-tile form (results wrong by construction)
cd "$GRAFT_REPO_ROOT"
for WMX in 2 4; do
  echo "== A3D_X3W_WM=$WMX"
  A3D_X3W_WM=$WMX python3 tools/x3w_check.py 2>&1 | cut -c1-150
done
for A in $ABLS; do
  echo "ABL=$A: $(A3D_X3W_WM=4 A3D_X3W_ABL=$A python3 tools/x3w_check.py 64x120x160x256x256 2>&1 | tail -1 | cut -c1-110)"
done
