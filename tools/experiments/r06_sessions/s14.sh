# all eight shares of the 64-set, whole path, for three group policies of the eager fitness searches
for G in 12 3 2; do
  for F in 0 8 16 24 32 40 48 56; do
    echo -n "group_min $G first $F: "
    LSR_FIT_GROUP_MIN=$G FIRST=$F MODE=share REPS=10 timeout 300 python tools/share_probe.py 2>&1 | tail -1
  done
done
