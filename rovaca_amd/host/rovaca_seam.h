// rovaca_seam.h -- the slice of Rovaca's domain types that the PairHMM seam touches, with the reference's names,
// signatures and semantics, so that pairhmm_gpu_engine.cpp reads (and compiles) the same inside the Rovaca tree.
//
// Inside Rovaca this header is NOT used: the adapter includes the real headers instead:
//     src/haplotypecaller/common/enum.h            PcrIndelModel                                   (:8)
//     src/haplotypecaller/genotype/forward.h       pMemoryPool, DoubleVector(2D), ReadVector, ...  (:21-96)
//     src/haplotypecaller/genotype/genotype_struct.h  Bases                                        (:11-15)
//     src/haplotypecaller/genotype/read_record.h   ReadRecord::{seq_length, decode_base, qual, mapping_quality}
//     src/haplotypecaller/genotype/allele.h        Haplotype (an Allele)::{get_display_string, length} (:104-107)
// Only the members the path uses are declared.  Storage here is plain vectors; the reference wraps a bam1_t.
#ifndef ROVACA_AMD_ROVACA_SEAM_H_
#define ROVACA_AMD_ROVACA_SEAM_H_
#include <cstdint>
#include <cstring>
#include <memory_resource>
#include <vector>

#ifdef ROVACA_AMD_WITH_REFERENCE_ENUM_AND_STRUCT
// Partial in-tree build (tests/test_capi_cpu.py): the two boost-free headers of the reference replace the mirror's enum and
// Bases, so that a drift between the mirror and the reference is a compile error here, not a link error in Rovaca.
#include "common/enum.h"
#include "genotype/genotype_struct.h"
#else
// GLOBAL namespace, as in the reference (common/enum.h:8): the enum is part of the seam's mangled signature.
enum class PcrIndelModel : uint8_t { NONE = 0, HOSTILE, AGGRESSIVE, CONSERVATIVE };
#endif

namespace rovaca
{

typedef std::pmr::memory_resource MemoryPool, *pMemoryPool;

#ifndef ROVACA_AMD_WITH_REFERENCE_ENUM_AND_STRUCT
typedef struct Bases  // genotype/genotype_struct.h:11-15
{
    uint32_t num;
    uint8_t data[];
} Bases, *pBases;
#else
typedef struct Bases Bases, *pBases;  // genotype/forward.h:24
#endif

// genotype/genotype_macors.h:36
#define ALLOC_FLEXIBLE_IN_POOL(pool, type1, num, type2) (pool->allocate(sizeof(type1) + sizeof(type2) * (num + 1)))

typedef class ReadRecord ReadRecord, *pReadRecord;
typedef class Haplotype Haplotype, *pHaplotype;
typedef std::pmr::vector<int32_t> Int32Vector;  // genotype/forward.h:75
typedef std::pmr::vector<double> DoubleVector;
typedef std::pmr::vector<DoubleVector> DoubleVector2D;
typedef std::pmr::vector<pReadRecord> ReadVector;
typedef std::pmr::vector<pHaplotype> HaplotypeVector;

class ReadRecord
{
public:
    ReadRecord(const uint8_t *bases, const uint8_t *quals, uint32_t len, uint8_t mapq)
        : bases_(bases, bases + len), quals_(quals, quals + len), mapq_(mapq)
    {}
    int32_t seq_length() const { return (int32_t)bases_.size(); }
    uint8_t mapping_quality() const { return mapq_; }
    uint8_t *qual() const { return const_cast<uint8_t *>(quals_.data()); }
    void decode_base(pBases base) const  // genotype/read_record.cpp:176-182
    {
        base->num = (uint32_t)bases_.size();
        memcpy(base->data, bases_.data(), bases_.size());
    }

private:
    std::vector<uint8_t> bases_, quals_;
    uint8_t mapq_;
};

class Haplotype
{
public:
    Haplotype(const uint8_t *bases, uint32_t len)
    {
        store_.resize(sizeof(Bases) + len + 1);
        pBases b = reinterpret_cast<pBases>(store_.data());
        b->num = len;
        memcpy(b->data, bases, len);
    }
    pBases get_display_string() const { return reinterpret_cast<pBases>(const_cast<uint8_t *>(store_.data())); }
    uint32_t length() const { return get_display_string()->num; }

private:
    std::vector<uint8_t> store_;
};

}  // namespace rovaca
#endif
