"""Regenerates gettysburg.txt and kats.json from DATA values held in the reference's sources/tests.

Authoring-container only (needs /root/reference); the GPU box never runs this.  Only data values
(test input string, constant tables, expected outputs) are extracted -- no reference code.
"""
import json
import re

R = "/root/reference"


def main():
    src = open(f"{R}/primitives/tests/polynomial_test.rs", encoding="utf-8").read()
    text = re.search(r'GETTYSBURG_ADDRESS_BYTES: &\[u8\] = "(.*?)"\.as_bytes\(\)', src, re.S).group(1)
    open("gettysburg.txt", "w", encoding="utf-8").write(text)
    c = open(f"{R}/primitives/src/consts.rs").read()
    blk = c[c.index("PRIMITIVE_ROOTS_OF_UNITY"):c.index("G2_TAU")]
    roots = re.findall(r'MontFp!\("(\d+)"\)', blk)
    assert len(roots) == 29
    kats = json.load(open("kats.json"))       # montgomery_reduce / pad_payload vectors were typed in by hand
    kats["primitive_roots_of_unity"] = roots
    json.dump(kats, open("kats.json", "w"), indent=1)


if __name__ == "__main__":
    main()
