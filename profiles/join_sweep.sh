cd /root/repo
export SF_DEBUG_KNOBS=1
for fo in "12000 3000" "12000 6000" "8000 3000" "16000 6000" "6000 1500" "20000 3000"; do
  set -- $fo
  echo "floor $1 ovh $2"; SF_JOIN_FLOOR=$1 SF_JOIN_OVH=$2 timeout 200 python profiles/join_probe2.py 2>&1 | head -1
done
