# A/B two builds of the library on the kernel micro-benchmark: bash tools/ab.sh libA.so libB.so
cd $GRAFT_REPO_ROOT
for lib in "$@"; do
  echo "== $lib"
  TIKE_AMD_LIB=$PWD/tike_amd/csrc/$lib python3 tools/kbench.py --det 256 --tiles 8000 --modes 8 --c3 | grep -v "\.\.\|copy"
done
