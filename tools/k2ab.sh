R=$GRAFT_REPO_ROOT; cd $R
V=$R/g-nerf_amd/gnerf_hip/variants
for lib in default ksplit ksplit_nofence; do
  if [ $lib = default ]; then unset GNERF_HIP_LIB; else export GNERF_HIP_LIB=$V/libgnerf_$lib.so; fi
  echo "== $lib"
  BWD_ONLY=staged python tools/bench_bwd.py 4 128 2>/dev/null | cut -c1-200
  python tools/dbg_bwd_det.py 2>/dev/null | cut -c1-90
done
unset GNERF_HIP_LIB
echo "== default, K2 forced f32"
GNERF_BWD_MLP_K2=f32 BWD_ONLY=staged python tools/bench_bwd.py 4 128 2>/dev/null | cut -c1-200
