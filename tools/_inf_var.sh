cd $GRAFT_REPO_ROOT
for v in 234 216; do
  echo "== tok $v"; SPX_INFLATE_TOK=$v timeout 300 python3 -m pytest tests/test_inflate.py -x -q -m gpu 2>&1 | tail -4
done
for v in 232 233 234 235 216; do
  echo "== tok $v stage 1"
  SPX_INFLATE_FLAT=1000 SPX_INFLATE_TOK_STAGE=1 SPX_INFLATE_TOK=$v timeout 300 python3 tools/inflate_bench.py --groups 49152 2>&1 | tail -1
done
echo "== tok 234 stage 3"
SPX_INFLATE_TOK=234 timeout 300 python3 tools/inflate_bench.py --groups 49152 2>&1 | tail -1
