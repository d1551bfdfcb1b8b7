"""Variants of a SNP-list VCF that drive the other branches of the reference's VCF pass (src/qv.cc:1642-1745): input that
already declares GT, input with FORMAT + sample columns, CHROM written with and without the "chr" prefix, blank lines.
Pure functions of the input text: tests/golden/make_golden.py feeds them to the reference binary and commits what it
wrote (tests/golden/ftiny.out.<kind>.vcf.gz); tests/test_host_tools.py feeds them to the product."""

GT_DECL = '##FORMAT=<ID=GT,Number=1,Type=String,Description="Genotype">\n'
KINDS = ("gtdecl", "fmtcols", "gtsecond")


def make(text, kind):
    out = []
    n = 0
    for ln in text.splitlines(keepends=True):
        if ln.startswith("##"):
            out.append(ln)
        elif ln.startswith("#"):
            if kind in ("gtdecl", "gtsecond"):
                out.append(GT_DECL)
            out.append(ln.rstrip("\n") + "\tFORMAT\tS1\n")
        else:
            n += 1
            c = ln.rstrip("\n").split("\t")
            if n % 5 == 0:
                c[0] = "chr" + c[0]                         # the pass prepends "chr" only when CHROM does not start with 'c'
            if kind == "gtdecl":
                c += ["GT:DP", "./.:%d" % (n % 40)]
            elif kind == "gtsecond":
                c += ["DP:GT:XX", "%d:./.:a" % (n % 40)]
            else:
                c += ["DP", "%d" % (n % 40)]
            out.append("\t".join(c) + "\n")
            if n % 97 == 0:
                out.append("\n")                            # empty lines are skipped
    return "".join(out)


# ---- variants of the INFO column, for the INDEX side (dictgen.c:561-794): the reference keeps, from line to line, the place
# of the token after the last "CAF..." key it has seen (`freq_index`) and reads that token on records that have no CAF key of
# their own; and gives up on frequencies for good if the FIRST record it parses has none.  tests/golden/make_golden.py indexes
# them with the reference binary and commits the sha256 of its files (tests/golden/ftiny.info_<kind>.sha256).
INFO_KINDS = ("stale", "nofreq", "unknownchr")


def _interleave_by_chrom(text):
    """The data lines dealt out chromosome by chromosome in turn (the reference takes its SNP list in any order)."""
    head, by = [], {}
    for ln in text.splitlines(keepends=True):
        if ln.startswith("#"):
            head.append(ln)
        else:
            by.setdefault(ln.split("\t", 1)[0], []).append(ln)
    out, queues = [], [q[::-1] for q in by.values()]
    while any(queues):
        for q in queues:
            if q:
                out.append(q.pop())
    return "".join(head + out)


def info_variant(text, kind):
    out = []
    n = 0
    if kind == "unknownchr":                                 # neighbours from different chromosomes: the bit-vector pass reads a
        text = _interleave_by_chrom(text)                    # record with an unknown name against the sequence of the record before
    for ln in text.splitlines(keepends=True):
        if ln.startswith("#"):
            out.append(ln)
            continue
        n += 1
        c = ln.rstrip("\n").split("\t")
        caf = [kv for kv in c[7].split(";") if kv.startswith("CAF=")][0]       # the list has "RS=..;CAF=r,a;COMMON=1": token 3
        if kind == "nofreq":
            if n == 1:
                c[7] = "RS=1;VC=SNV"                         # no CAF key on the first record: every frequency becomes 0.5
        else:
            if n % 5 == 0:
                c[7] = caf                                   # CAF as the only key: the token after it is number 1
            elif n % 3 == 0:
                c[7] = "AA=0.125,0.5;BB=0.25,0.75"           # no CAF key: token 1 or token 3, whichever the last CAF record left
            # (a record with FEWER tokens than that place is left out: the reference then reads a pointer left over from an
            # earlier line, dictgen.c:538-553 -- whatever the line buffer holds there; the product defines 0.5 / 0.5)
            if kind == "unknownchr" and n % 11 in (0, 1, 2):
                c[0] = "scaffold_%d" % (n % 4)               # names the FASTA does not have, in runs
        out.append("\t".join(c) + "\n")
    return "".join(out)
