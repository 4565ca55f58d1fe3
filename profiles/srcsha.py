"""Fingerprint of the kernel sources (vi-hds_amd/csrc/*.hip, *.hpp, Makefile): written into every PMC reduction under
profiles/ by make_pmc_traffic.py / make_pmc_valu.py and compared by bench.py before it attaches a committed counter figure
to a kernel it has just timed -- a traffic file measured on other kernel code is reported as stale, not as `traffic`."""
import glob
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def csrc_sha16():
    h = hashlib.sha256()
    d = os.path.join(ROOT, "vi-hds_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(d, "*.hip")) + glob.glob(os.path.join(d, "*.hpp")) + [os.path.join(d, "Makefile")]):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(csrc_sha16())
