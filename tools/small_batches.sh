# developer measurement: K1p vs K1q on SMALL batches (1k ... 262k rays), kernel time; K1q spreads a small batch over all waves of the chip
export RAYS=1024,4096,16384,65536,131072,262144
echo "== hall D=64"; timeout -k 10 300 python tools/ab_pool.py persist:default pool:default
echo "== cathedral D=128"; SCENE=cathedral DOMAIN=128 timeout -k 10 400 python tools/ab_pool.py persist:default pool:default
