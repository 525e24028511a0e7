"""`python -m shrimp_amd.mergesam [options] <reads> <s1.sam> <s2.sam> ...` -- the reference's mergesam command line (ref: mergesam/mergesam.c:153-243)
over gm_merge_sam.  SAM goes to stdout; --un / --al write the FASTA / FASTQ text to the named file, as the reference does."""
import argparse, gzip, sys
from . import gmapper as gm


def _read(path):
    with open(path, "rb") as f:
        data = f.read()
    return gzip.decompress(data) if data[:2] == b"\x1f\x8b" else data


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    ap = argparse.ArgumentParser(prog="mergesam", add_help=True)
    ap.add_argument("--un"); ap.add_argument("--al")
    ap.add_argument("--sam-unaligned", action="store_true"); ap.add_argument("-o", "--report", type=int, default=10)
    ap.add_argument("-N", "--threads", type=int, default=1); ap.add_argument("-E", "--sam", action="store_true")
    ap.add_argument("-Q", "--fastq", action="store_true"); ap.add_argument("--strata", action="store_true")
    ap.add_argument("--max-alignments", type=int, default=0)
    ap.add_argument("--no-half-paired", action="store_true"); ap.add_argument("--half-paired", action="store_true")
    ap.add_argument("--insert-size-dist"); ap.add_argument("--single-best-mapping", action="store_true")
    ap.add_argument("--min-mapq", type=int, default=0); ap.add_argument("--all-contigs", action="store_true")
    ap.add_argument("--no-mapping-qualities", action="store_true"); ap.add_argument("--leave-mapq-untouched", action="store_true")
    ap.add_argument("--sam-header"); ap.add_argument("--no-improper-mappings", action="store_true"); ap.add_argument("--no-autodetect-input", action="store_true")
    for unused in ("--buffer-size", "--read-size", "--read-rate", "-s", "--stack-size"):      # the reference's streaming parameters: nothing to tune here
        ap.add_argument(unused)
    ap.add_argument("reads"); ap.add_argument("sams", nargs="+")
    a = ap.parse_args(argv)
    if a.un and a.al:
        sys.exit(" ! Please, '--un' xor '--al' == 1!")
    if not (a.sam or a.un or a.al):
        sys.exit(" ! Mergesam currently only supports output in SAM or FAST(A/Q) format, please use one of '--un','--al',or '--sam'")
    out = "un" if a.un else ("al" if a.al else "sam")
    text = gm.merge_sam(_read(a.reads), [_read(s) for s in a.sams], command_line=" ".join(["mergesam"] + argv) + " ", output=out,
                        max_outputs=a.report, max_alignments=a.max_alignments, strata=a.strata, half_paired=not a.no_half_paired,
                        sam_unaligned=a.sam_unaligned, single_best=a.single_best_mapping, all_contigs=a.all_contigs,
                        no_mapping_qualities=a.no_mapping_qualities, leave_mapq=a.leave_mapq_untouched, no_improper_mappings=a.no_improper_mappings,
                        min_mapq=a.min_mapq, fastq=(1 if a.fastq else (0 if a.no_autodetect_input else -1)), threads=a.threads,
                        header_given=bool(a.sam_header))
    if a.sam_header:
        h = _read(a.sam_header); sys.stdout.buffer.write(h if h.endswith(b"\n") else h + b"\n")
    if out == "sam":
        sys.stdout.buffer.write(text)
    else:
        with open(a.un or a.al, "wb") as f:
            f.write(text)
    return 0


if __name__ == "__main__":
    sys.exit(main())
