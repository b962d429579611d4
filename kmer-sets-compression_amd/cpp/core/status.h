// Minimal Status / StatusOr with the surface the reference uses from abseil
// (absl::Status, absl::StatusOr, OkStatus, InternalError, FailedPreconditionError;
// lib/core/io.h:26,43,63, lib/core/kmer_set_set.h:465,525,610).  abseil is not in
// this image; inside the reference tree these map 1:1 onto the absl types.
#ifndef KSC_CORE_STATUS_H_
#define KSC_CORE_STATUS_H_

#include <string>
#include <utility>

#include "kmersets_hip.h"

namespace ksc {

enum class StatusCode { kOk = 0, kInvalidArgument = 3, kFailedPrecondition = 9, kInternal = 13 };

class Status {
 public:
  Status() = default;
  Status(StatusCode code, std::string message) : code_(code), message_(std::move(message)) {}
  bool ok() const { return code_ == StatusCode::kOk; }
  StatusCode code() const { return code_; }
  const std::string& message() const { return message_; }
  std::string ToString() const {
    if (ok()) return "OK";
    const char* name = code_ == StatusCode::kInternal             ? "INTERNAL"
                       : code_ == StatusCode::kFailedPrecondition ? "FAILED_PRECONDITION"
                                                                  : "INVALID_ARGUMENT";
    return std::string(name) + ": " + message_;
  }

 private:
  StatusCode code_ = StatusCode::kOk;
  std::string message_;
};

inline Status OkStatus() { return Status(); }
inline Status InternalError(std::string m) { return Status(StatusCode::kInternal, std::move(m)); }
inline Status FailedPreconditionError(std::string m) {
  return Status(StatusCode::kFailedPrecondition, std::move(m));
}

// Status of a C-ABI return code (the message is the library's thread-local one).
inline Status FromKsh(int rc) {
  if (rc == KSH_OK) return Status();
  return Status(static_cast<StatusCode>(rc), ksh_last_error());
}

template <typename T>
class StatusOr {
 public:
  StatusOr(Status s) : status_(std::move(s)) {}           // NOLINT
  StatusOr(T value) : value_(std::move(value)) {}         // NOLINT
  bool ok() const { return status_.ok(); }
  const Status& status() const { return status_; }
  T& value() & { return value_; }
  const T& value() const& { return value_; }
  T&& value() && { return std::move(value_); }

 private:
  Status status_;
  T value_{};
};

}  // namespace ksc

#endif
