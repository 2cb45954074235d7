cd ${GRAFT_REPO_ROOT:-.}
python -m pytest tests/test_gpu_parity.py tests/test_known_answers.py tests/test_gpu_training.py tests/test_gpu_e2e.py -q -m gpu -k "rpn or nms or proposal or select or stage or matcher or sampl or known or matched" 2>&1 | tail -3
bash tools/probes/r06_b1prof.sh 2>&1 | grep -E "group_nms|rpn_select|nms_mask|bench line|roi_align|paste"
python bench.py --no-cpu-baseline --no-train-leg --no-alt-modes --no-operating-points --steps 20 2>/dev/null > /tmp/b.json; python -c "import json;d=json.loads(open('/tmp/b.json').read().splitlines()[0]);print('bench64', d['value'], d['ms_per_step'])"
