for v in 5 3 2 5 3 2; do NOVA_HIP_SHARED_BLOCKS_PER_CU=$v timeout -k 10 200 python3 scripts/gpu_shared_percu.py 2>&1 | grep shared_per_cu || exit 1; done
