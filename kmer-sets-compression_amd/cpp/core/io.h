// Text lines in and out, optionally through an external (de)compressor -- the on-disk
// side of the reference (lib/core/io.h:20-126): one string per line, '\n' terminated;
// "<decompressor> < file" / "<compressor> > file" through popen; same error messages.
#ifndef KSC_CORE_IO_H_
#define KSC_CORE_IO_H_

#include <cstdio>
#include <fstream>
#include <string>
#include <vector>

#include "core/status.h"

namespace ksc {

inline StatusOr<std::vector<std::string>> ReadLines(const std::string& file_name,
                                                    const std::string& decompressor) {
  std::vector<std::string> lines;
  if (decompressor.empty()) {
    std::ifstream file(file_name);
    if (file.fail()) return InternalError("failed to open file");
    std::string s;
    while (std::getline(file, s)) lines.push_back(s);
    return lines;
  }
  std::FILE* f = popen((decompressor + " < " + file_name).c_str(), "r");
  if (f == NULL) return InternalError("failed to open a sub-process");
  std::string s;
  {
    char buf[8192];
    while (fgets(buf, sizeof(buf), f) != NULL) s += buf;
  }
  const int exit_status = pclose(f);
  if (exit_status != 0)
    return InternalError("process failed with non-zero exit code: " + std::to_string(exit_status));
  if (!s.empty() && s.back() == '\n') s.pop_back();
  // absl::StrSplit(s, '\n'): an empty input yields one empty piece (io.h:68-70)
  std::size_t at = 0;
  while (true) {
    const std::size_t nl = s.find('\n', at);
    if (nl == std::string::npos) {
      lines.push_back(s.substr(at));
      break;
    }
    lines.push_back(s.substr(at, nl - at));
    at = nl + 1;
  }
  return lines;
}

inline Status WriteLines(const std::string& file_name, const std::string& compressor,
                         const std::vector<std::string>& lines) {
  if (compressor.empty()) {
    std::ofstream file(file_name);
    if (file.fail()) return InternalError("failed to open file");
    for (const std::string& line : lines) file << line << '\n';
    return OkStatus();
  }
  std::FILE* f = popen((compressor + " > " + file_name).c_str(), "w");
  if (f == NULL) return InternalError("failed to open a sub-process");
  for (const std::string& line : lines) {
    if (std::fputs(line.c_str(), f) == EOF || std::fputc('\n', f) == EOF)
      return InternalError("failed to write to the process");
  }
  const int exit_status = pclose(f);
  if (exit_status != 0)
    return InternalError("process failed with non-zero exit code: " + std::to_string(exit_status));
  return OkStatus();
}

// The same files as whole byte buffers: the device spells and parses the lines
// (ksh_spss_to_text / ksh_spss_from_text_*), the host only moves bytes.
inline StatusOr<std::string> ReadBytes(const std::string& file_name, const std::string& decompressor) {
  std::string s;
  if (decompressor.empty()) {
    std::ifstream file(file_name, std::ios::binary);
    if (file.fail()) return InternalError("failed to open file");
    file.seekg(0, std::ios::end);
    const std::streamoff size = file.tellg();
    file.seekg(0, std::ios::beg);
    s.resize(static_cast<std::size_t>(size));
    if (size > 0) file.read(&s[0], size);
    if (file.fail()) return InternalError("failed to read file");
    return s;
  }
  std::FILE* f = popen((decompressor + " < " + file_name).c_str(), "r");
  if (f == NULL) return InternalError("failed to open a sub-process");
  {
    char buf[1 << 16];
    std::size_t got;
    while ((got = std::fread(buf, 1, sizeof(buf), f)) > 0) s.append(buf, got);
  }
  const int exit_status = pclose(f);
  if (exit_status != 0)
    return InternalError("process failed with non-zero exit code: " + std::to_string(exit_status));
  return s;
}

inline Status WriteBytes(const std::string& file_name, const std::string& compressor, const char* data,
                         std::size_t size) {
  if (compressor.empty()) {
    std::ofstream file(file_name, std::ios::binary);
    if (file.fail()) return InternalError("failed to open file");
    file.write(data, static_cast<std::streamsize>(size));
    if (file.fail()) return InternalError("failed to write file");
    return OkStatus();
  }
  std::FILE* f = popen((compressor + " > " + file_name).c_str(), "w");
  if (f == NULL) return InternalError("failed to open a sub-process");
  if (size > 0 && std::fwrite(data, 1, size, f) != size) {
    pclose(f);
    return InternalError("failed to write to the process");
  }
  const int exit_status = pclose(f);
  if (exit_status != 0)
    return InternalError("process failed with non-zero exit code: " + std::to_string(exit_status));
  return OkStatus();
}

}  // namespace ksc

#endif
