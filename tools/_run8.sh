cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_train_gpu.py -x -q -k "gradient_goldens" -s 2>&1 | tail -12
python tools/train_glue_trace.py 8 split 2>/dev/null | tail -50
