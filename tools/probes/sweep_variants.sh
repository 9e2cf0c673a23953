# usage: sweep_variants.sh "<time_step.py arguments>" variant...   (libbase9hip.so and every build/variants/lib_<variant>.so)
ARGS=$1; shift
python3 tools/time_step.py $ARGS 2>&1 | grep "us/step"
for v in "$@"; do B9_HIP_LIB=build/variants/lib_$v.so python3 tools/time_step.py $ARGS 2>&1 | grep "us/step"; done
