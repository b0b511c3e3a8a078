cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
cd $R && timeout -k 10 900 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -2
timeout -k 10 200 python3 $R/tools/full_model_profile.py --R 5 --steps 10 2>/dev/null | tail -1
