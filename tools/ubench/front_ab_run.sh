for args in "8192" "8192" "2048 101 64 2 1" "512 61 40 0 0" "512 101 30 1 0" "300 101 64 0 1" "64 33 96 0 0" "64 200 64 0 0"; do
  echo "== $args"; tools/ubench/front_ab_before $args | tail -2; tools/ubench/front_ab $args | tail -2
done
