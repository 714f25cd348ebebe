#!/usr/bin/env python3
"""sha256 of the machine code of every kernel in libuchirp.so (per gfx950 code object, per kernel symbol: the bytes
[st_value, st_value + st_size) of .text) -- the tripwire behind "this change did not touch that kernel":
  python tools/kernel_digest.py                 print {kernel: [digest per code object]}
  python tools/kernel_digest.py --update        rewrite tests/golden/kernel_digests.json from the library as built
  python tools/kernel_digest.py --diff          names whose digests differ from the recorded ones
tests/test_kernel_resources.py compares the shipped library with the record (no GPU needed)."""
import hashlib
import json
import os
import shutil
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.environ.get("UCHIRP_LIB") or os.path.join(ROOT, "ultrasonic-communication_amd", "libuchirp.so")
RECORD = os.path.join(ROOT, "tests", "golden", "kernel_digests.json")
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def _elf_kernels(path):
    """{symbol: sha256 of its bytes} for every FUNC symbol of an ELF64 little-endian code object"""
    with open(path, "rb") as f:
        b = f.read()
    assert b[:4] == b"\x7fELF" and b[4] == 2 and b[5] == 1, path
    shoff, = struct.unpack_from("<Q", b, 0x28)
    shentsize, shnum, _ = struct.unpack_from("<HHH", b, 0x3A)
    sh = [struct.unpack_from("<IIQQQQIIQQ", b, shoff + i * shentsize) for i in range(shnum)]
    out = {}
    for (_, typ, _, _, off, size, link, _, _, entsize) in sh:
        if typ != 2:  # SHT_SYMTAB
            continue
        stroff = sh[link][4]
        for i in range(size // entsize):
            name, info, _, shndx, value, ssize = struct.unpack_from("<IBBHQQ", b, off + i * entsize)
            if (info & 0xF) != 2 or ssize == 0 or shndx == 0 or shndx >= shnum:  # STT_FUNC, defined
                continue
            sec = sh[shndx]
            start = sec[4] + (value - sec[3])
            end = b.index(b"\0", stroff + name)
            out[b[stroff + name:end].decode()] = hashlib.sha256(b[start:start + ssize]).hexdigest()
    return out


def digests(lib=LIB):
    """{kernel: sorted list of digests, one per code object that holds it}"""
    work = tempfile.mkdtemp(prefix="ucdig")
    try:
        shutil.copy(lib, os.path.join(work, "lib.so"))
        subprocess.run([OBJDUMP, "--offloading", "lib.so"], check=True, cwd=work, stdout=subprocess.DEVNULL,
                       stderr=subprocess.DEVNULL)
        out = {}
        for f in sorted(os.listdir(work)):
            if "amdgcn" not in f:
                continue
            for k, d in _elf_kernels(os.path.join(work, f)).items():
                out.setdefault(k, []).append(d)
        return {k: sorted(v) for k, v in sorted(out.items())}
    finally:
        shutil.rmtree(work, ignore_errors=True)


def main():
    d = digests()
    if "--update" in sys.argv:
        with open(RECORD, "w") as f:
            json.dump(d, f, indent=0, sort_keys=True)
            f.write("\n")
        print("recorded %d kernels in %s" % (len(d), RECORD))
    elif "--diff" in sys.argv:
        with open(RECORD) as f:
            old = json.load(f)
        for k in sorted(set(old) | set(d)):
            if old.get(k) != d.get(k):
                print(("changed " if k in old and k in d else "added   " if k in d else "removed ") + k)
    else:
        json.dump(d, sys.stdout, indent=0, sort_keys=True)


if __name__ == "__main__":
    main()
