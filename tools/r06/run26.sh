python -m pytest tests/test_gpu_models.py -x -q -k "segmented or sibling_pools or paired_skinny or grouped" 2>&1 | tail -6
python -m pytest tests/test_gpu_plan_api.py tests/test_gpu_speed2d.py -x -q 2>&1 | tail -6
