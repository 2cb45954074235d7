# timing-only ablations of wino_gemm_x3w_kernel<4> (library built with A3D_HIPCC_FLAGS=-DA3D_ABLATIONS; results wrong by construction)
cd "$GRAFT_REPO_ROOT"
for A in $VARS; do
  echo "VAR=$A: $(A3D_X3W_VAR=$A python3 tools/x3w_check.py 64x120x160x256x256 2>&1 | tail -1 | cut -c1-110)"
done
