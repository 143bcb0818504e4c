// ReorderBase — static one-liner facade (reference: bases/reorder_base.h:29-710).
// Signatures follow the reference so call sites compile unchanged.
#ifndef SPARSEBASE_BASES_REORDER_BASE_H_
#define SPARSEBASE_BASES_REORDER_BASE_H_
#include "sparsebase/permute/permute_order_one.h"
#include "sparsebase/permute/permute_order_two.h"
#include "sparsebase/reorder/degree_reorder.h"
#include "sparsebase/reorder/generic_reorder.h"
#include "sparsebase/reorder/gray_reorder.h"
#include "sparsebase/reorder/rcm_reorder.h"

namespace sparsebase::bases {

class ReorderBase {
  template <typename I, typename N, typename V>
  using F2 = format::FormatOrderTwo<I, N, V>;

  template <template <typename, typename, typename> typename Ret, typename I, typename N, typename V>
  static Ret<I, N, V> *Finish(F2<I, N, V> *out, bool convert_output) {
    if constexpr (std::is_same_v<Ret<I, N, V>, F2<I, N, V>>) return out;
    else if (convert_output) return out->template Convert<Ret>();
    else return out->template As<Ret>();
  }
  template <typename I, typename N, typename V>
  static std::vector<F2<I, N, V> *> Cast(const std::vector<format::Format *> &in) {
    std::vector<F2<I, N, V> *> out;
    for (auto *f : in) out.push_back(static_cast<F2<I, N, V> *>(f));
    return out;
  }

 public:
  template <template <typename, typename, typename> typename Reordering, typename I, typename N, typename V>
  static I *Reorder(typename Reordering<I, N, V>::ParamsType params, F2<I, N, V> *format,
                    std::vector<context::Context *> contexts, bool convert_input) {
    static_assert(std::is_base_of_v<reorder::Reorderer<I>, Reordering<I, N, V>>,
                  "You must pass a reordering function (with base Reorderer) to ReorderBase::Reorder");
    static_assert(!std::is_same_v<reorder::GenericReorder<I, N, V>, Reordering<I, N, V>>,
                  "GenericReorder cannot be used through ReorderBase::Reorder");
    Reordering<I, N, V> reordering(params);
    return reordering.GetReorder(format, contexts, convert_input);
  }
  // Device-resident form (additive; the context is taken by reference so that the reference's `{&context}` call sites
  // keep selecting the overload above): the order vector comes back as an HIPArray<I> in the HBM of `context`'s device and
  // goes into the Permute2D overloads below as it is — the canonical pipeline (experiment/experiment_helper.h:81-97)
  // without the order vector's two trips over PCIe.  The caller owns the array.
  template <template <typename, typename, typename> typename Reordering, typename I, typename N, typename V>
  static format::HIPArray<I> *Reorder(typename Reordering<I, N, V>::ParamsType params, F2<I, N, V> *format,
                                      context::HIPContext &context, bool convert_input = true) {
    static_assert(std::is_base_of_v<reorder::Reorderer<I>, Reordering<I, N, V>>,
                  "You must pass a reordering function (with base Reorderer) to ReorderBase::Reorder");
    Reordering<I, N, V> reordering(params);
    return reordering.GetReorderDevice(format, &context, convert_input);
  }
  template <template <typename, typename, typename> typename Ret = format::FormatOrderTwo, typename I, typename N,
            typename V>
  static Ret<I, N, V> *Permute2D(format::HIPArray<I> *ordering, F2<I, N, V> *format,
                                 std::vector<context::Context *> contexts, bool convert_input,
                                 bool convert_output = false) {
    return Permute2DRowColumnWise<Ret>(ordering, ordering, format, contexts, convert_input, convert_output);
  }
  template <template <typename, typename, typename> typename Ret = format::FormatOrderTwo, typename I, typename N,
            typename V>
  static Ret<I, N, V> *Permute2DRowColumnWise(format::HIPArray<I> *row_ordering, format::HIPArray<I> *col_ordering,
                                              F2<I, N, V> *format, std::vector<context::Context *> contexts,
                                              bool convert_input, bool convert_output = false) {
    permute::PermuteOrderTwo<I, N, V> perm(row_ordering, col_ordering);
    return Finish<Ret>(perm.GetPermutation(format, contexts, convert_input), convert_output);
  }
  template <template <typename, typename, typename> typename Ret = format::FormatOrderTwo, typename I, typename N,
            typename V>
  static Ret<I, N, V> *Permute2DRowWise(format::HIPArray<I> *ordering, F2<I, N, V> *format,
                                        std::vector<context::Context *> contexts, bool convert_input,
                                        bool convert_output = false) {
    return Permute2DRowColumnWise<Ret>(ordering, (format::HIPArray<I> *)nullptr, format, contexts, convert_input,
                                       convert_output);
  }
  template <template <typename, typename, typename> typename Ret = format::FormatOrderTwo, typename I, typename N,
            typename V>
  static Ret<I, N, V> *Permute2DColWise(format::HIPArray<I> *ordering, F2<I, N, V> *format,
                                        std::vector<context::Context *> contexts, bool convert_input,
                                        bool convert_output = false) {
    return Permute2DRowColumnWise<Ret>((format::HIPArray<I> *)nullptr, ordering, format, contexts, convert_input,
                                       convert_output);
  }
  template <template <typename, typename, typename> typename Reordering, typename I, typename N, typename V>
  static std::pair<std::vector<F2<I, N, V> *>, I *> ReorderCached(typename Reordering<I, N, V>::ParamsType params,
                                                                  F2<I, N, V> *format,
                                                                  std::vector<context::Context *> contexts) {
    Reordering<I, N, V> reordering(params);
    auto out = reordering.GetReorderCached(format, contexts, true);
    return std::make_pair(Cast<I, N, V>(std::get<0>(out)[0]), std::get<1>(out));
  }

  template <template <typename, typename, typename> typename Ret = format::FormatOrderTwo, typename I, typename N,
            typename V>
  static Ret<I, N, V> *Permute2D(I *ordering, F2<I, N, V> *format, std::vector<context::Context *> contexts,
                                 bool convert_input, bool convert_output = false) {
    return Permute2DRowColumnWise<Ret>(ordering, ordering, format, contexts, convert_input, convert_output);
  }
  // Permute2D over the GPUs of a node, one process per GPU (no reference counterpart): every rank holds
  // the same HIPCSR and ordering, permutes its own new-row range and gets the whole row_ptr by all-gather.
  template <typename I, typename N, typename V>
  static permute::ShardedHIPCSR<I, N, V> *Permute2DSharded(I *ordering, format::HIPCSR<I, N, V> *format,
                                                           context::HIPCommunicator &comm,
                                                           const int64_t *row_splits = nullptr) {
    return Permute2DRowColumnWiseSharded(ordering, ordering, format, comm, row_splits);
  }
  template <typename I, typename N, typename V>
  static permute::ShardedHIPCSR<I, N, V> *Permute2DRowColumnWiseSharded(I *row_ordering, I *col_ordering,
                                                                        format::HIPCSR<I, N, V> *format,
                                                                        context::HIPCommunicator &comm,
                                                                        const int64_t *row_splits = nullptr) {
    permute::PermuteOrderTwo<I, N, V> perm(row_ordering, col_ordering);
    return perm.GetPermutationSharded(format, comm, row_splits);
  }
  template <template <typename, typename, typename> typename Ret = format::FormatOrderTwo, typename I, typename N,
            typename V>
  static std::pair<std::vector<F2<I, N, V> *>, Ret<I, N, V> *> Permute2DCached(
      I *ordering, F2<I, N, V> *format, std::vector<context::Context *> contexts, bool convert_output = false) {
    return Permute2DRowColumnWiseCached<Ret>(ordering, ordering, format, contexts, convert_output);
  }
  template <template <typename, typename, typename> typename Ret = format::FormatOrderTwo, typename I, typename N,
            typename V>
  static Ret<I, N, V> *Permute2DRowColumnWise(I *row_ordering, I *col_ordering, F2<I, N, V> *format,
                                              std::vector<context::Context *> contexts, bool convert_input,
                                              bool convert_output = false) {
    permute::PermuteOrderTwo<I, N, V> perm(row_ordering, col_ordering);
    return Finish<Ret>(perm.GetPermutation(format, contexts, convert_input), convert_output);
  }
  template <template <typename, typename, typename> typename Ret = format::FormatOrderTwo, typename I, typename N,
            typename V>
  static std::pair<std::vector<F2<I, N, V> *>, Ret<I, N, V> *> Permute2DRowColumnWiseCached(
      I *row_ordering, I *col_ordering, F2<I, N, V> *format, std::vector<context::Context *> contexts,
      bool convert_output = false) {
    permute::PermuteOrderTwo<I, N, V> perm(row_ordering, col_ordering);
    auto out = perm.GetPermutationCached(format, contexts, true);
    return std::make_pair(Cast<I, N, V>(std::get<0>(out)[0]), Finish<Ret>(std::get<1>(out), convert_output));
  }
  template <template <typename, typename, typename> typename Ret = format::FormatOrderTwo, typename I, typename N,
            typename V>
  static Ret<I, N, V> *Permute2DRowWise(I *ordering, F2<I, N, V> *format, std::vector<context::Context *> contexts,
                                        bool convert_input, bool convert_output = false) {
    return Permute2DRowColumnWise<Ret>(ordering, (I *)nullptr, format, contexts, convert_input, convert_output);
  }
  template <template <typename, typename, typename> typename Ret = format::FormatOrderTwo, typename I, typename N,
            typename V>
  static std::pair<std::vector<F2<I, N, V> *>, Ret<I, N, V> *> Permute2DRowWiseCached(
      I *ordering, F2<I, N, V> *format, std::vector<context::Context *> contexts, bool convert_output = false) {
    return Permute2DRowColumnWiseCached<Ret>(ordering, (I *)nullptr, format, contexts, convert_output);
  }
  template <template <typename, typename, typename> typename Ret = format::FormatOrderTwo, typename I, typename N,
            typename V>
  static Ret<I, N, V> *Permute2DColWise(I *ordering, F2<I, N, V> *format, std::vector<context::Context *> contexts,
                                        bool convert_input, bool convert_output = false) {
    return Permute2DRowColumnWise<Ret>((I *)nullptr, ordering, format, contexts, convert_input, convert_output);
  }
  template <template <typename, typename, typename> typename Ret = format::FormatOrderTwo, typename I, typename N,
            typename V>
  static std::pair<std::vector<F2<I, N, V> *>, Ret<I, N, V> *> Permute2DColWiseCached(
      I *ordering, F2<I, N, V> *format, std::vector<context::Context *> contexts, bool convert_output = false) {
    return Permute2DRowColumnWiseCached<Ret>((I *)nullptr, ordering, format, contexts, convert_output);
  }

  template <template <typename> typename Ret = format::FormatOrderOne, typename I, typename V>
  static Ret<V> *Permute1D(I *ordering, format::FormatOrderOne<V> *format, std::vector<context::Context *> contexts,
                           bool convert_inputs, bool convert_output = false) {
    permute::PermuteOrderOne<I, V> perm(ordering);
    auto *out = perm.GetPermutation(format, contexts, convert_inputs);
    if constexpr (std::is_same_v<Ret<V>, format::FormatOrderOne<V>>) return out;
    else if (convert_output) return out->template Convert<Ret>();
    else return out->template As<Ret>();
  }

  // inv[perm[i]] = i on the GPU (reference :663-672); result is a host new[] array
  template <typename I, typename NumElements>
  static I *InversePermutation(I *perm, NumElements length) {
    static_assert(std::is_integral_v<NumElements>, "Length of the permutation array must be an integer");
    auto &dev = hip::Device::Get(hip::DefaultDevice());
    hip::Staged<I> d_perm(dev, perm, (size_t)length), d_inv(dev, (size_t)(length ? length : 1));
    dev.Check(sbx_inverse_permutation(dev.handle(), hip::IndexTag<I>(), (int64_t)length, d_perm.get(), d_inv.get()));
    I *inv = new I[length];
    if (length) dev.ToHost(inv, d_inv.get(), (size_t)length * sizeof(I));
    return inv;
  }
};

}  // namespace sparsebase::bases
#endif
