cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 600 python -m pytest tests -m gpu -q -x --tb=short -p no:cacheprovider -k "alternative or backbone_vs" 2>&1 | tail -3
python tools/ab_option.py block1 4 5 2>&1 | grep -v amdgpu.ids | tail -6
