set -u
R=$GRAFT_REPO_ROOT; O=gpurun_out/trace50k; mkdir -p $R/$O; cd /tmp; export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $R/$O/t -o t -- python $R/tools/hosttime.py 50000 > $R/$O/log.txt 2>&1
db=$(find $R/$O/t -name "*.db" | head -1)
python $R/tools/rocpd_summary.py $db --out $R/$O/summary.md --title "B=50000" | tail -1
cat $R/$O/summary.md | head -20
python $R/tools/rocpd_timeline.py $db 2>/dev/null | tail -40
find $R/$O -name "*.db" -size +4M -delete
