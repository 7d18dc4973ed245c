for t in 16 32 64 128 256; do echo "ticket $t"; HARE_TICKET=$t timeout -k 10 200 python tools/ab_pool.py pool:hare_amd/libhare_hip_rf64.so pool:hare_amd/libhare_hip_st192.so pool:hare_amd/libhare_hip_st256.so; done
export HARE_DEV=1   # developer overrides (HARE_VOXEL_KERNEL, HARE_TICKET, ...) are only read in a process that opted in
