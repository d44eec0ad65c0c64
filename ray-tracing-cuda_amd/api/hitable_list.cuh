#pragma once
#include "hitable.cuh"

// HitableList: fixed array of up to 1024 entries, appended in order (hitable_list.cuh:10-21).
class HitableList : public Hitable {
 public:
  constexpr static int kMaxHitables = 1024;
  RT_API HitableList() : Hitable(rtapi::H_LIST) {}
  RT_API void Append(Hitable *obj) { list_[list_len_++] = obj; }
  RT_API int list_len() const { return list_len_; }
  RT_API Hitable *at(int i) const { return list_[i]; }

 private:
  Hitable *list_[kMaxHitables];
  int list_len_ = 0;
};
