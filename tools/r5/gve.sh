cd $GRAFT_REPO_ROOT
for i in 1 2; do echo "== eager2 native"; python tools/graph_vs_eager.py eager2 2>&1 | grep -v amdgpu | tail -3; done
echo "== graph native"; python tools/graph_vs_eager.py graph 2>&1 | grep -v amdgpu | tail -3
for i in 1 2; do echo "== eager2 blas"; SN_FWD_BLAS=1 python tools/graph_vs_eager.py eager2 2>&1 | grep -v amdgpu | tail -3; done
python -m pytest tests/test_gpu_round5.py -q -m gpu -k "native_fp32" 2>&1 | tail -2
