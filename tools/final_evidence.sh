: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "$0")/.." && pwd)}"
# The evidence of a round from ONE box: smoke, the full GPU test run on the default flavour, the default bench line with the
# rocprofv3 summaries / timeline / PMC traffic (tools/refresh_profiles.sh), the distributed lines (tools/dist_lines.sh).
#   /usr/local/graft/bin/gpurun --timeout 1200 -- 'bash tools/final_evidence.sh'
cd $GRAFT_REPO_ROOT
R=${R:-r06}
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${R}_smoke.log 2>&1; tail -1 gpurun_out/${R}_smoke.log
timeout -k 10 700 python -m pytest tests -m gpu -q > gpurun_out/${R}_gputests_default.log 2>&1; tail -2 gpurun_out/${R}_gputests_default.log
R=$R bash tools/refresh_profiles.sh > gpurun_out/${R}_refresh.log 2>&1; tail -c 300 gpurun_out/${R}_bench_default.json; echo
R=$R bash tools/dist_lines.sh > gpurun_out/${R}_dist_lines.log 2>&1; tail -c 200 gpurun_out/${R}_bench_rccl1.json; echo
