cd /root/repo
for i in 1 2 3; do python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-extra 2>&1 | grep ^{ | python -c "import sys,json; print('default', json.loads(sys.stdin.read())['ms_per_step'])"; SEFD_TUNING=CG256_MINTILES=130 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-extra 2>&1 | grep ^{ | python -c "import sys,json; print('mintiles130', json.loads(sys.stdin.read())['ms_per_step'])"; done
python tools/opbench.py --tags 201 --ab "" "CG256_MINTILES=130" 2>&1 | tail -14
