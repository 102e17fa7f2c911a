// Host mirror of smvs::ConjugateGradient and smvs::BlockSparseMatrix<4>
// (reference: lib/conjugate_gradient.h:19-56, lib/block_sparse_matrix.h:30-98)
// with the reference's signatures; the solve runs on the GPU through
// smvs_gn_upload / smvs_cg_solve of include/smvs_hip.h.
//
// This is the compatibility surface of SURVEY.md 8(b) rows 6-7: a caller that
// still drives construct() and solve() itself (the reference's
// run_newton_iterations, lib/depth_optimizer.cc:225-262) links against it
// unchanged.  It moves the whole system across PCIe on every call; the fast
// path is DepthOptimizer -> smvs_gn_run_loop, which never materialises H.
#pragma once

#include <cstddef>
#include <vector>

struct smvs_ctx;

namespace smvs_amd {

class ConjugateGradient
{
public:
    typedef std::vector<double> Vector;   // SSEVector (lib/sse_vector.h:21-56)

    enum ReturnInfo   // lib/conjugate_gradient.h:22-27
    {
        CG_CONVERGENCE,
        CG_MAX_ITERATIONS,
        CG_INVALID_INPUT
    };

    struct Options    // :29-35
    {
        int max_iterations = 1000;
        double error_tolerance = 1e-20;
        double q_tolerance = 1e-3;
    };

    struct Status     // :37-42
    {
        int num_iterations = 0;
        ReturnInfo info = CG_INVALID_INPUT;
    };

    // :44-50, the reference's interface member for member.  The reference's
    // solver only calls multiply(); the device solver needs the matrix itself,
    // so solve() accepts the one Functor whose matrix it can see:
    // BlockStencilMatrix (what GaussNewtonStep::construct produces) -- any
    // other Functor is refused with std::invalid_argument, not silently solved
    // on the host.  multiply() is there for the callers that use the operator
    // outside solve() (residual checks: H x - b).
    class Functor
    {
    public:
        virtual ~Functor(void) {}
        virtual Vector multiply(Vector const& x) const = 0;
        virtual std::size_t input_size(void) const = 0;
        virtual std::size_t output_size(void) const = 0;
    };

public:
    explicit ConjugateGradient(Options const& opts, int device = 0);
    ~ConjugateGradient(void);
    ConjugateGradient(ConjugateGradient const&) = delete;
    ConjugateGradient& operator=(ConjugateGradient const&) = delete;

    // :55-56.  Throws std::invalid_argument on a dimension mismatch like the
    // reference (:77-78) and when A / P are not BlockStencilMatrix.
    Status solve(Functor const& A, Vector const& b, Vector* x,
        Functor const* P = nullptr);

private:
    Options opts;
    Status status;
    int device;
    smvs_ctx* ctx = nullptr;
    std::size_t ctx_nodes = 0, ctx_stride = 0;
};

// BlockSparseMatrix<4> of a surface's node grid in block-stencil form: block
// (row node n, column node n + dy * stride + dx) at blocks[(n * 9 + s) * 16],
// s = (dy + 1) * 3 + dx + 1, 4 x 4 row-major, zero where the reference holds
// no block (gauss_newton_step.cc:99-121 only creates blocks between nodes that
// share a patch).  A preconditioner has the diagonal slot s = 4 only.
class BlockStencilMatrix : public ConjugateGradient::Functor
{
public:
    std::size_t num_nodes = 0, node_stride = 0;
    std::vector<double> blocks;

    // BlockSparseMatrix<4>::multiply (block_sparse_matrix.h:276-298) on the
    // host: per output row the products in the reference's order (ascending
    // block column, then column inside the block, one multiplication and one
    // addition each); std::invalid_argument on a size mismatch (:279-280)
    ConjugateGradient::Vector multiply(ConjugateGradient::Vector const& x) const;
    std::size_t input_size(void) const { return 4 * num_nodes; }
    std::size_t output_size(void) const { return 4 * num_nodes; }
    std::size_t num_non_zero(void) const;   // block_sparse_matrix.h:70 (entries)
};

} // namespace smvs_amd
