bash tools/exp_ab.sh
