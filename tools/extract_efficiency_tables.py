#!/usr/bin/env python3
"""Extracts the tabulated collision efficiencies (numeric DATA: published efficiencies of Hall 1980,
Davis 1972 / Jonas 1972, Pinsky et al. 2008, Vohl et al. 2007 on the reference's (R, r) grid) from the
reference checkout into small binary files the library loads at run time:

    libcloudphxx_amd/data/kernel_eff_<kernel_t value>.f64  =  [r_max, count, count x float64]

Only numbers are taken; no code is copied.  Run once in the build container (the reference is not
available on the GPU box):   python tools/extract_efficiency_tables.py [/root/reference]
Layout of a table: lower-triangular matrix, entry (i, j<=i) at i(i+1)/2 + j with i, j = kernel_index(radius in um)
(reference: src/detail/kernel_utils.hpp:10-30, src/detail/kernel_interpolation.hpp:9-65).
"""
import os
import re
import struct
import sys

KERNELS = {"hall": 3, "hall_davis_no_waals": 4, "hall_pinsky_1000mb_grav": 8, "hall_pinsky_cumulonimbus": 9,
           "hall_pinsky_stratocumulus": 10, "vohl_davis_no_waals": 11}


def main():
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "libcloudphxx_amd", "data")
    os.makedirs(out_dir, exist_ok=True)
    for name, kid in KERNELS.items():
        src = open(os.path.join(ref, "src", "detail", "kernel_definitions", name + "_efficiencies.hpp")).read()
        r_max = float(re.search(r"_r_max\(\)\s*\{\s*return\s*([0-9.eE+-]+)\s*;", src).group(1))
        body = src[src.index("arr[] = {") + len("arr[] = {"):]
        body = body[:body.index("};")]
        vals = [float(t) for t in re.findall(r"[-+]?\d*\.?\d+(?:[eE][-+]?\d+)?", body)]
        with open(os.path.join(out_dir, "kernel_eff_%d.f64" % kid), "wb") as f:
            f.write(struct.pack("<2d", r_max, float(len(vals))))
            f.write(struct.pack("<%dd" % len(vals), *vals))
        print(name, kid, "r_max", r_max, "entries", len(vals))


if __name__ == "__main__":
    main()
