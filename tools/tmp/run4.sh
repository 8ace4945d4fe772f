cd $GRAFT_REPO_ROOT
timeout 300 python3 -m pytest tests/test_inflate.py -x -q -m gpu 2>&1 | tail -2
echo "== stage 3"; timeout 300 python3 tools/inflate_bench.py --groups 49152 2>&1 | tail -1
timeout 300 python3 tools/fuzz_inflate.py 17 45 2>&1 | tail -1
