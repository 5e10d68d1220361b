#!/bin/bash
# every opcode of op_rates.hip in its own process (a hung kernel costs 10 s, not the run); usage: op_rates_all.sh > table.txt
cd "$(dirname "$0")"
for id in $(seq 0 56) $(seq 60 64) $(seq 70 76); do
  timeout 10 ./op_rates $id || echo "op $id: no result (timeout or fault)"
done
