# developer: is the bounce loop (C5) bound by where the polygon records live?  Same 8-bounce loop on a scene whose records are
export HARE_DEV=1   # developer overrides (HARE_VOXEL_KERNEL, HARE_TICKET, ...) are only read in a process that opted in
# L2 / Infinity-Cache resident (hall, 12.9 MB) and on the cathedral (126 MB); bench.py computes B/cast from the oracle on the
# same rays, so the two `roofline.frac` values compare per algorithmic byte.  One step after another; nothing is retried.
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 280 python3 bench.py --scene hall --domain 64 --bounces 8 --steps 5 --warmup 1 --no-e2e > gpurun_out/r2_b8_hall.json 2> gpurun_out/r2_b8_hall.err &&
timeout -k 10 280 python3 bench.py --scene cathedral --domain 128 --bounces 8 --steps 5 --warmup 1 --no-e2e > gpurun_out/r2_b8_cath.json 2> gpurun_out/r2_b8_cath.err
rc=$?
tail -n 2 gpurun_out/r2_b8_hall.json gpurun_out/r2_b8_cath.json
tail -n 3 gpurun_out/r2_b8_hall.err gpurun_out/r2_b8_cath.err
exit $rc
