# same-box A/B of the configs[4] step over library builds:  VARIANTS="base _name ..." bash scripts/ab_config4.sh   (`make variant NAME=name`)
for i in 1 2 3; do for v in ${VARIANTS:-base}; do
  [ "$v" = base ] && s="" || s="$v"
  echo -n "lib$s config4: "
  VD_LIB_PATH=$PWD/visdial_amd/libvisdial_hip$s.so timeout 200 python bench.py --config ${CONFIG:-4} --steps 30 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])"
done; done
