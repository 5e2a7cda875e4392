for d in 0 2 3 4 5; do echo "== CGH_PB_DEBUG=$d"; CGH_PB_DEBUG=$d bash tools/prof_pb.sh 5 pbd$d | grep -E "pb_expand|pb_accum|layout" || exit 1; done
