for b in 0 96 224; do for t in 0 448; do
echo "== block $b tid $t"; SKHIP_STACK_STAMP_BLOCK=$b SKHIP_STACK_STAMP_TID=$t timeout -k 10 100 python tools/stack_stamps.py 2>/dev/null | tail -17 || exit 1
done; done
