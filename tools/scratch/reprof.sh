cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
export O=gpurun_out
rm -rf $O/prof_final $O/pmc_fetch_f $O/pmc_write_f
sed -n '11,13p' tools/refresh_profiles.sh > /tmp/_p.sh
bash /tmp/_p.sh
tail -c 300 $O/bench_final_prof.json; echo
