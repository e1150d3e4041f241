#!/bin/bash
# builds and runs cpu_syrk_probe.cpp with every ISA candidate of oracle.py's build sweep (+ two that separate the ISA from
# the tuning) on the host at hand; output: what `native` resolves to, then one block per flag set
cd "$(dirname "$0")" || exit 1
echo "host: $(grep -m1 'model name' /proc/cpuinfo | cut -d: -f2)  gcc $(gcc -dumpversion)"
echo "native resolves to: $(gcc -march=native -Q --help=target 2>/dev/null | grep -E '^\s+-(march|mtune)=' | tr -s ' ' | tr '\n' ' ')"
for f in "-march=x86-64-v3" "-march=x86-64-v4" "-march=native" "-march=native -mtune=generic" "-march=native -mno-avx512f" "-march=x86-64-v4 -mprefer-vector-width=512"; do
  echo "== $f"
  g++ -O3 -std=c++17 $f -I../../include -o /tmp/cpu_syrk_probe cpu_syrk_probe.cpp -pthread 2>&1 | head -3 && /tmp/cpu_syrk_probe
done
