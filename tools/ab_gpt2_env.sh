run() { name=$1; shift; env "$@" python bench.py --config gpt2 --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', round(d['ms_per_step'],2), round(d['roofline']['decode_ms_per_population'],2))"; }
run base X=1
run kernarg1 HIP_FORCE_DEV_KERNARG=1
run kernarg0 HIP_FORCE_DEV_KERNARG=0
run pktcap DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run pktcap0 DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run nograph GLASS_GPT2_NO_GRAPH=1
run base2 X=1
