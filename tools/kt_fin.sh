cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
export JDA_LANES=1 JDA_SIDE_STREAM=0 VAR_STEPS=10
for tw in ${TWS:-0 57 71}; do
  rm -rf /tmp/kt; JDA_FIN_TILE=${tw} JDA_FIN_GRID_DIV=${GD:-4} JDA_FIN_LDS_EXTRA=${EX:-0} rocprofv3 --kernel-trace --stats -d /tmp/kt -- python $R/tools/variants.py "" > /dev/null 2>&1
  echo "== JDA_FIN_TILE=$tw"; python $R/tools/rocpd_summary.py $(find /tmp/kt -name "*.db" | head -1) k_finish | grep -E "^[0-9]+ .*k_finish" | cut -c1-110
done
