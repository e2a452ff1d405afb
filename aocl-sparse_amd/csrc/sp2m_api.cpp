// sp2m_api.cpp -- aoclsparse_sp2m / aoclsparse_spmm / aoclsparse_?csr2m entry points.
//
// Argument checks follow level3/aoclsparse_csr2m.cpp:592-740 of the reference.  The sparse x sparse
// product itself (two-stage Gustavson, csr2m.cpp:46-543) has no HIP kernel yet: rather than ship a CPU
// loop inside the GPU product, valid requests return aoclsparse_status_not_implemented (DESIGN.md,
// "open rows").  The oracle restates the algorithm (oracle.c: orc_csr2m_nnz / orc_dcsr2m_fill).
#include "internal.hpp"

using namespace mi355;

namespace
{
aoclsparse_status sp2m_checks(aoclsparse_operation opA, const aoclsparse_mat_descr descrA,
                              const aoclsparse_matrix A, aoclsparse_operation opB,
                              const aoclsparse_mat_descr descrB, const aoclsparse_matrix B,
                              const aoclsparse_request request, aoclsparse_matrix *C)
{
    if(!A || !B || !C || !descrA || !descrB)
        return aoclsparse_status_invalid_pointer;
    if(!A->user.ptr || !B->user.ptr)
        return aoclsparse_status_invalid_pointer;
    if(A->val_type != B->val_type)
        return aoclsparse_status_wrong_type;
    if(descrA->base != A->base || descrB->base != B->base)
        return aoclsparse_status_invalid_value;
    auto valid_op = [](aoclsparse_operation o) {
        return o == aoclsparse_operation_none || o == aoclsparse_operation_transpose
               || o == aoclsparse_operation_conjugate_transpose;
    };
    if(!valid_op(opA) || !valid_op(opB))
        return aoclsparse_status_invalid_value;
    if(request != aoclsparse_stage_nnz_count && request != aoclsparse_stage_finalize
       && request != aoclsparse_stage_full_computation)
        return aoclsparse_status_invalid_value;
    if(descrA->type != aoclsparse_matrix_type_general || descrB->type != aoclsparse_matrix_type_general)
        return aoclsparse_status_not_implemented;
    // inner dimensions of op(A) * op(B)
    const aoclsparse_int ka = opA == aoclsparse_operation_none ? A->n : A->m;
    const aoclsparse_int kb = opB == aoclsparse_operation_none ? B->m : B->n;
    if(ka != kb)
        return aoclsparse_status_invalid_size;
    return aoclsparse_status_not_implemented;
}
} // namespace

extern "C" {

aoclsparse_status aoclsparse_sp2m(aoclsparse_operation opA, const aoclsparse_mat_descr descrA,
                                  const aoclsparse_matrix A, aoclsparse_operation opB,
                                  const aoclsparse_mat_descr descrB, const aoclsparse_matrix B,
                                  const aoclsparse_request request, aoclsparse_matrix *C)
{
    return sp2m_checks(opA, descrA, A, opB, descrB, B, request, C);
}

aoclsparse_status aoclsparse_spmm(aoclsparse_operation opA, const aoclsparse_matrix A,
                                  const aoclsparse_matrix B, aoclsparse_matrix *C)
{
    // level3/aoclsparse_spmm.cpp:27-67: general descriptors in each matrix's base, full computation
    if(!A || !B || !C)
        return aoclsparse_status_invalid_pointer;
    _aoclsparse_mat_descr dA, dB;
    dA.base = A->base;
    dB.base = B->base;
    return sp2m_checks(opA, &dA, A, aoclsparse_operation_none, &dB, B, aoclsparse_stage_full_computation, C);
}

aoclsparse_status aoclsparse_dcsr2m(aoclsparse_operation trans_A, const aoclsparse_mat_descr descrA,
                                    const aoclsparse_matrix csrA, aoclsparse_operation trans_B,
                                    const aoclsparse_mat_descr descrB, const aoclsparse_matrix csrB,
                                    const aoclsparse_request request, aoclsparse_matrix *csrC)
{
    if(csrA && csrA->val_type != aoclsparse_dmat)
        return aoclsparse_status_wrong_type;
    return sp2m_checks(trans_A, descrA, csrA, trans_B, descrB, csrB, request, csrC);
}

aoclsparse_status aoclsparse_scsr2m(aoclsparse_operation trans_A, const aoclsparse_mat_descr descrA,
                                    const aoclsparse_matrix csrA, aoclsparse_operation trans_B,
                                    const aoclsparse_mat_descr descrB, const aoclsparse_matrix csrB,
                                    const aoclsparse_request request, aoclsparse_matrix *csrC)
{
    if(csrA && csrA->val_type != aoclsparse_smat)
        return aoclsparse_status_wrong_type;
    return sp2m_checks(trans_A, descrA, csrA, trans_B, descrB, csrB, request, csrC);
}

} // extern "C"
