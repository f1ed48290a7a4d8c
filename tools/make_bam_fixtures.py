"""Copies the DATA of reference testcases that come with a BAM into tests/golden/bam/<name>/ (run here, where /root/reference exists;
the fixtures travel, the reference does not): the BAM and ref.fa verbatim, scenario.yaml verbatim, and variant.tsv = the
CHROM/POS/ID/REF/ALT columns of the first record of candidates.vcf.  The `expected:` block of each testcase.yaml is restated as a
predicate in tests/bam_pairs.py:BAM_CASES (with the tests/lib.rs line of the testcase).

    python tools/make_bam_fixtures.py test_giab_05 test_giab_12 ...
"""
import glob
import os
import shutil
import sys

SRC = "/root/reference/tests/resources/testcases"
DST = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "bam")

for name in sys.argv[1:]:
    s, d = os.path.join(SRC, name), os.path.join(DST, name)
    os.makedirs(d, exist_ok=True)
    bam, = glob.glob(os.path.join(s, "*.bam"))
    for f in (bam, os.path.join(s, "ref.fa"), os.path.join(s, "scenario.yaml")):
        shutil.copyfile(f, os.path.join(d, os.path.basename(f)))
    rec = [l for l in open(os.path.join(s, "candidates.vcf")) if not l.startswith("#")][0].rstrip("\n").split("\t")[:5]
    with open(os.path.join(d, "variant.tsv"), "w") as out:
        out.write("#CHROM\tPOS\tID\tREF\tALT\n" + "\t".join(rec) + "\n")
    print(name, rec, os.path.basename(bam))
