cd $GRAFT_REPO_ROOT
for lib in libtike_amd.so libtike_amd_g2.so libtike_amd_g4.so; do
  echo "== $lib"
  TIKE_AMD_LIB=$PWD/tike_amd/csrc/$lib python3 tools/kbench.py --det 256 --tiles 8000 --modes 8 --c3 | grep "grad_ifft"
done
