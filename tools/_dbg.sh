cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for d in 1 2 3 4 8 16 64; do
rm -rf /tmp/ktD
PATS_PK_GRID=$d B=4096 C=264 NTOK=145 N=2 rocprofv3 --kernel-trace --stats -d /tmp/ktD -- python $R/tools/pmc_gnn.py > /dev/null 2>&1
echo "dbg=$d"; python $R/tools/rocpd_stats.py $(find /tmp/ktD -name "*.db" | head -1) "x" --list conv_pk | grep '^- ' | head -6 | awk '{print $4}' | tr '\n' ' '; echo
done
