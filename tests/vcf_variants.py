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
