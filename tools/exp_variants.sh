for v in "" SAMEADDR NOLOAD NOLDS; do
  if [ -n "$v" ]; then export DSAT_LIB=$PWD/tools/libs/libdsat_$v.so; else unset DSAT_LIB; fi
  echo "== variant ${v:-base}"
  for cfg in "1 1 1 8" "2 1 4 8" "3 1 1 8"; do echo -n "mlp/impl/rb/nw $cfg: "; timeout -k 5 120 python tests/tools/mlp_one.py $cfg 2>/dev/null | tail -1; done
done
