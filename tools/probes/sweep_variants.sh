# usage: sweep_variants.sh "<shapes>" variant...   (time_step.py --marg 4 4 with libbase9hip.so and every build/variants/lib_<variant>.so)
SHAPES=$1; shift
python3 tools/time_step.py $SHAPES --marg 4 4 2>&1 | grep "us/step"
for v in "$@"; do B9_HIP_LIB=build/variants/lib_$v.so python3 tools/time_step.py $SHAPES --marg 4 4 2>&1 | grep "us/step"; done
