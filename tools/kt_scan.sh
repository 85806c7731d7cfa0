cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
export JDA_LANES=1 JDA_SIDE_STREAM=0 VAR_STEPS=10
for pl in ${PLS:-1 2 0}; do
  rm -rf /tmp/kt; JDA_SCAN_PLANES=$pl JDA_TILES=${TL:-} rocprofv3 --kernel-trace --stats -d /tmp/kt -- python $R/tools/variants.py "" > /dev/null 2>&1
  echo "== JDA_SCAN_PLANES=$pl"; python $R/tools/rocpd_summary.py $(find /tmp/kt -name "*.db" | head -1) k_scan | grep -E "^ +[0-9.]+ us" | head -5 | cut -c1-90
done
