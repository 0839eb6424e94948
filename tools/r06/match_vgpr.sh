R=${GRAFT_REPO_ROOT:-$PWD}
for rnd in 1 2 3; do for v in cur m_vgpr2; do
 if [ "$v" = "cur" ]; then unset HESS_LIB; else export HESS_LIB=$R/tools/_variants/$v/libhessgpu.so; fi
 echo "$v $(python tools/bench_match.py | grep -o '"n1": [0-9]*, "n2": [0-9]*, "device_ms": [0-9.]*, "GMAC_per_s": [0-9.]*' | tail -2 | tr '\n' ' ')"
done; done
