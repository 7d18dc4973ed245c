# developer measurement: K1p vs K1q over batch sizes on three scenes (where choose_kernel's crossover comes from); kernel time, CRC-equal events
export RAYS=131072,262144,393216,524288,786432,1048576,1572864
echo "== hall D=64"; timeout -k 10 300 python tools/ab_pool.py persist:default pool:default
echo "== hall D=128"; DOMAIN=128 timeout -k 10 300 python tools/ab_pool.py persist:default pool:default
echo "== cathedral D=128"; SCENE=cathedral DOMAIN=128 RAYS=131072,262144,393216,524288,1048576 timeout -k 10 400 python tools/ab_pool.py persist:default pool:default
