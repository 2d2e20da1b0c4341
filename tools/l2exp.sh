cd $GRAFT_REPO_ROOT
for lib in libtike_amd.so libtike_amd_a.so libtike_amd_b.so; do
  echo "== $lib"
  TIKE_AMD_LIB=$PWD/tike_amd/csrc/$lib python3 tools/kbench.py --det 256 --tiles 8000 --modes 8 --c3 | grep -v "\.\."
done
