#!/usr/bin/env python3
"""Clock and socket power of one GPU, sampled from its sysfs directory by a process of its own (bench.py starts it: a thread inside the
benchmark would take the interpreter lock from the loop that enqueues the kernels).

usage: gpu_state_sampler.py <sysfs device dir | auto> <out file> [period_ms=2]
One line per sample: <CLOCK_MONOTONIC ns> <sclk MHz> <power W>  (-1 where a value cannot be read).  Runs until stdin reaches end of file."""
import glob
import os
import select
import sys
import time


def device_dir(arg):
    if arg != "auto" and os.path.isdir(arg):
        return arg
    for card in sorted(glob.glob("/sys/class/drm/card*/device")):
        if glob.glob(os.path.join(card, "hwmon/hwmon*/power1_*")) or os.path.isfile(os.path.join(card, "pp_dpm_sclk")):
            return card
    return None


def main():
    card = device_dir(sys.argv[1])
    period = (float(sys.argv[3]) if len(sys.argv) > 3 else 2.0) / 1e3
    power = None
    if card:
        for pat in ("hwmon/hwmon*/power1_average", "hwmon/hwmon*/power1_input"):
            hits = glob.glob(os.path.join(card, pat))
            if hits:
                power = hits[0]
                break
    sclk = os.path.join(card, "pp_dpm_sclk") if card else None
    with open(sys.argv[2], "w") as out:
        out.write("# %s\n" % (card or "no amdgpu device directory found"))
        out.flush()
        while not select.select([sys.stdin], [], [], 0)[0] or sys.stdin.readline():
            t = time.monotonic_ns()
            mhz = watts = -1.0
            try:
                for line in open(sclk).read().splitlines():
                    if line.rstrip().endswith("*"):
                        mhz = float("".join(ch for ch in line.split(":")[1] if ch.isdigit() or ch == "."))
            except (OSError, ValueError, TypeError, IndexError):
                pass
            try:
                watts = int(open(power).read()) / 1e6
            except (OSError, ValueError, TypeError):
                pass
            out.write("%d %.0f %.1f\n" % (t, mhz, watts))
            out.flush()
            time.sleep(period)


if __name__ == "__main__":
    main()
