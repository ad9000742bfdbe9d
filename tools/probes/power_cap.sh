#!/bin/bash
# Prints the package power cap and clock limits of the box (sysfs hwmon + rocm-smi), for DESIGN §4.13.
for h in /sys/class/drm/card*/device/hwmon/hwmon*; do
  echo "== $h"
  for f in power1_cap power1_cap_max power1_cap_min power1_cap_default power1_average power1_input freq1_input; do
    [ -r $h/$f ] && echo "$f $(cat $h/$f)"
  done
done
rocm-smi --showmaxpower --showpower --showclocks --showperflevel 2>&1 | grep -v "^$" | head -40
for d in /sys/class/drm/card*/device; do [ -r $d/pp_dpm_sclk ] && { echo "== $d/pp_dpm_sclk"; cat $d/pp_dpm_sclk; }; done
