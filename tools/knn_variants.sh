# Diagnostic (GPU box): time kernel variants built beforehand into build/var/libpsg_<name>.so
#   (make -C pointsecguard_amd/csrc EXTRA="..." && cp pointsecguard_amd/libpsg.so build/var/libpsg_<name>.so)
# names ending in "s": stamp builds (-DPSG_KF_STAMP), names starting with "T": timeline builds (-DPSG_KF_TL)
for v in ${VARIANTS:?names}; do cp build/var/libpsg_$v.so pointsecguard_amd/libpsg.so; echo "variant $v"
  case $v in T*) timeout -k 10 100 python tools/knn_timeline.py 4 || exit 1;; *s) timeout -k 10 100 python tools/knn_stamp.py || exit 1;; *) timeout -k 10 100 python tools/knn_time.py 4 || exit 1;; esac; done
