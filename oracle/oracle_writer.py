"""ORACLE (test infrastructure only): restatement of GenotypeWriter's line assembly (src/bayesTyper/GenotypeWriter.cpp) around the
genotype-derived columns, which the C++ oracle produces (oracle_gibbs.cpp: orc_cluster_output_columns).  Pure Python: string
assembly over a handful of variants.  Parity unpinned (GenotypeWriter.cpp needs Boost iostreams; no reference fixture)."""


def vcf_header(genome_filename, chromosomes, graph_options_header, genotype_options_header, sample_names):
    """generateHeader (:494-551); chromosomes: [(name, sequence, is_decoy)] in genome order"""
    h = "##fileformat=VCFv4.2\n"
    h += "##reference=file:" + genome_filename + "\n"
    for name, seq, is_decoy in chromosomes:
        if not is_decoy:
            h += "##contig=<ID=%s,length=%d>\n" % (name, len(seq))
    h += graph_options_header
    h += genotype_options_header
    h += '##FILTER=<ID=AN0,Description="No called genotypes (AN = 0)">\n'
    h += '##INFO=<ID=AC,Number=A,Type=Integer,Description="Alternative allele counts in called genotypes">\n'
    h += '##INFO=<ID=AF,Number=A,Type=Float,Description="Alternative allele frequencies in called genotypes">\n'
    h += '##INFO=<ID=AN,Number=1,Type=Integer,Description="Total number of alleles in called genotypes">\n'
    h += '##INFO=<ID=ACP,Number=R,Type=Float,Description="Allele call probabilites (maximum APP across samples)">\n'
    h += '##INFO=<ID=VCS,Number=1,Type=Integer,Description="Variant cluster size">\n'
    h += '##INFO=<ID=VCR,Number=1,Type=String,Description="Variant cluster region (<chromosome>:<start>-<end>)">\n'
    h += '##INFO=<ID=VCGS,Number=1,Type=Integer,Description="Variant cluster group size (number of variant clusters)">\n'
    h += '##INFO=<ID=VCGR,Number=1,Type=String,Description="Variant cluster group region (<chromosome>:<start>-<end>)">\n'
    h += '##INFO=<ID=HC,Number=1,Type=Integer,Description="Number of haplotype candidates used for inference in variant cluster">\n'
    h += "##INFO=<ID=ANC,Number=.,Type=String,Description=\"Allele(s) not covered by a haplotype candidate ('0': Reference allele)\">\n"
    h += '##INFO=<ID=ACO,Number=A,Type=String,Description="Alternative allele call-set origin(s) (<call-set>:...)">\n'
    h += '##FORMAT=<ID=GT,Number=1,Type=String,Description="Genotype">\n'
    h += '##FORMAT=<ID=GQ,Number=1,Type=Integer,Description="Genotype quality (phred-scaled 1 - max(GPP))">\n'
    h += '##FORMAT=<ID=GPP,Number=G,Type=Float,Description="Genotype posterior probabilities">\n'
    h += '##FORMAT=<ID=APP,Number=R,Type=Float,Description="Allele posterior probabilities">\n'
    h += "##FORMAT=<ID=NAK,Number=R,Type=Float,Description=\"Mean number of allele kmers across gibbs samples ('-1': Not sampled)\">\n"
    h += "##FORMAT=<ID=FAK,Number=R,Type=Float,Description=\"Mean fraction of observed allele kmers across gibbs samples ('-1': Not sampled or NAK = 0)\">\n"
    h += "##FORMAT=<ID=MAC,Number=R,Type=Float,Description=\"Mean allele kmer coverage (mean value) across gibbs samples ('-1': Not sampled or NAK = 0)\">\n"
    h += "##FORMAT=<ID=SAF,Number=R,Type=Integer,Description=\"Sample specific allele filter ('0': PASS, '1': NAK, '2': FAK, '3': NAK and FAK)\">\n"
    h += "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT"
    for s in sample_names:
        h += "\t" + s
    return h + "\n"


def vcf_line(chrom_name, chrom_sequence, pos0, variant_id, has_dependency, alts, columns, vcs, vcr, vcgs, vcgr, hc):
    """one output line.  pos0: 0-based position; alts: [(ref_length, sequence, aco_att)]; columns: the oracle's
    "<QUAL>\\t<FILTER>\\tAC=..;ACP=..[;ANC=..]\\t<sample columns>" for the variant (writeGenotypes :84-128, finalise :452-470)"""
    position = pos0 + 1
    max_ref_length = max(a[0] for a in alts)                                     # VariantInfo::maxReferenceLength
    alt_field = ",".join(seq + chrom_sequence[position + rl - 1:position + rl - 1 + (max_ref_length - rl)] for rl, seq, _ in alts)   # writeAlleleSequences :145-172
    if has_dependency:
        alt_field += ",*"
    qual, filt, info, *samples = columns.split("\t")
    anc = ""
    if ";ANC=" in info:                                                          # writeAlleleCover comes after the cluster annotations
        info, anc = info.split(";ANC=")
        anc = ";ANC=" + anc
    info += ";VCS=%d;VCR=%s;VCGS=%d;VCGR=%s;HC=%d" % (vcs, vcr, vcgs, vcgr, hc) + anc
    info += ";ACO=" + ",".join(a[2] if a[2] else "." for a in alts) + (",." if has_dependency else "")   # writeAlleleOrigin :232-259
    ref = chrom_sequence[position - 1:position - 1 + max_ref_length]
    return "\t".join([chrom_name, str(position), variant_id, ref, alt_field, qual, filt, info, "GT:GQ:GPP:APP:NAK:FAK:MAC:SAF"] + samples) + "\n"
