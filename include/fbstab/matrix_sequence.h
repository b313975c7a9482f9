// MatrixSequence / MapMatrixSequence with the reference's memory layout
// (tools/matrix_sequence.h:18-164): element (i, j) of the k-th matrix lives at
// data[k*nrows*ncols + j*nrows + i].  operator()(k) returns a light view
// instead of an Eigen::Map; everything else (constructors, accessors, the
// exceptions thrown) mirrors the reference.
#pragma once

#include <stdexcept>
#include <vector>

namespace fbstab {

// Non-owning view of one matrix of a sequence (column-major).
template <class T>
class MatrixViewT {
 public:
  MatrixViewT(T* data, int rows, int cols) : data_(data), r_(rows), c_(cols) {}
  T& operator()(int i, int j) const { return data_[i + static_cast<size_t>(j) * r_]; }
  T& operator()(int i) const { return data_[i]; }
  int rows() const { return r_; }
  int cols() const { return c_; }
  int size() const { return r_ * c_; }
  T* data() const { return data_; }
  // copy from any object with rows()/cols()/operator()(i,j)
  template <class M>
  const MatrixViewT& operator=(const M& m) const {
    if (m.rows() != r_ || m.cols() != c_) throw std::runtime_error("MatrixView: size mismatch");
    for (int j = 0; j < c_; j++)
      for (int i = 0; i < r_; i++) (*this)(i, j) = m(i, j);
    return *this;
  }

 private:
  T* data_;
  int r_, c_;
};
typedef MatrixViewT<double> MatrixView;
typedef MatrixViewT<const double> ConstMatrixView;

class MatrixSequence {
 public:
  MatrixSequence() : N_(0), nr_(1), nc_(1), nel_(0) {}
  // tools/matrix_sequence.h:30-42
  MatrixSequence(int len, int nrows, int ncols = 1) {
    if (len < 0) throw std::runtime_error("Negative length input in MatrixSequence");
    if (nrows <= 0 || ncols <= 0)
      throw std::runtime_error("Non-positive row or column count in MatrixSequence");
    N_ = len;
    nr_ = nrows;
    nc_ = ncols;
    nel_ = N_ * nr_ * nc_;
    data_.resize(nel_);
  }
  MatrixView operator()(int k) {
    if (k < 0 || k >= N_) throw std::out_of_range("Bad indexing in MatrixSequence");
    return MatrixView(data() + static_cast<size_t>(k) * nr_ * nc_, nr_, nc_);
  }
  ConstMatrixView operator()(int k) const {
    if (k < 0 || k >= N_) throw std::out_of_range("Bad indexing in MatrixSequence");
    return ConstMatrixView(data() + static_cast<size_t>(k) * nr_ * nc_, nr_, nc_);
  }
  int rows() const { return nr_; }
  int cols() const { return nc_; }
  int length() const { return N_; }
  int size() const { return nel_; }
  double* data() { return data_.data(); }
  const double* data() const { return data_.data(); }

 private:
  int N_, nr_, nc_, nel_;
  std::vector<double> data_;
};

class MapMatrixSequence {
 public:
  MapMatrixSequence() : data_(nullptr), N_(0), nr_(1), nc_(1), nel_(0) {}
  // tools/matrix_sequence.h:103-121
  MapMatrixSequence(const double* data, int len, int nrows, int ncols) : data_(data) {
    if (len <= 0) throw std::runtime_error("Non-positive length input in MapMatrixSequence");
    if (nrows <= 0 || ncols <= 0)
      throw std::runtime_error("Non-positive row or column count in MapMatrixSequence");
    if (data == nullptr)
      throw std::runtime_error("Cannot initialize MapMatrixSequence will a nullptr");
    N_ = len;
    nr_ = nrows;
    nc_ = ncols;
    nel_ = N_ * nr_ * nc_;
  }
  MapMatrixSequence(const MatrixSequence& A)  // NOLINT: implicit like the reference (:129)
      : data_(A.data()), N_(A.length()), nr_(A.rows()), nc_(A.cols()), nel_(A.size()) {}
  ConstMatrixView operator()(int k) const {
    if (k < 0 || k >= N_) throw std::out_of_range("Bad indexing in MapMatrixSequence");
    if (data_ == nullptr) throw std::runtime_error("In MapMatrixSequence, cannot index into null data.");
    return ConstMatrixView(data_ + static_cast<size_t>(k) * nr_ * nc_, nr_, nc_);
  }
  int rows() const { return nr_; }
  int cols() const { return nc_; }
  int length() const { return N_; }
  int size() const { return nel_; }
  const double* data() const { return data_; }

 private:
  const double* data_;
  int N_, nr_, nc_, nel_;
};

}  // namespace fbstab
