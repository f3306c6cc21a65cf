// id3_tag.h -- carrying the source's ID3v2 tag over to the converted file.
//
// The reference copies ID3v2 tags "where possible" when the source is a .dsf or .dff that has them
// (README.md:7), appends " [<OUTPUT_RATE>]" to the album tag with -a (README.md:170-173,
// src/main.rs:118-124) and copies artwork files next to the outputs with -p (README.md:115-119,
// src/main.rs:41-47).  Its tag code lives in the absent rdsd2pcm crate (which uses the `id3` crate,
// Cargo.lock), so the details below are this build's own choices, stated where they are made:
//   * DSF: the tag is the blob at the header's metadata pointer; DFF: the payload of an 'ID3 ' chunk.
//     A tag that claims more bytes than the file holds (id3_test/*brokenid3*) is dropped with a
//     warning and the audio is converted regardless.
//   * WAV gets the tag as an 'id3 ' RIFF chunk after 'data', AIFF/AIFC as an 'ID3 ' chunk after
//     'SSND', FLAC as a VORBIS_COMMENT block (text frames mapped to the usual field names) plus one
//     PICTURE block per APIC frame.
//   * the album suffix is the abbreviated rate with a dot, e.g. " [88.2K]", " [96K]".
#pragma once
#include <stdint.h>

#include <string>
#include <utility>
#include <vector>

#include "dsd_reader.h"

namespace d2dhost {

// The complete ID3v2 tag (10-byte header + frames, padding dropped) of a DSF/DFF source, or empty.
// Returns "" or an I/O error; a damaged or missing tag is not an error (see `warning`).
std::string read_source_tag(const std::string& path, const DsdInfo& info, std::vector<uint8_t>& tag, std::string& warning);

// Appends `suffix` to the text of the album frame (TALB, or TAL in ID3v2.2) in the frame's own
// encoding.  Returns false (tag untouched) when there is no album frame or the tag uses features
// this editor does not rewrite (unsynchronisation, extended header, compressed/encrypted frame).
bool append_to_album(std::vector<uint8_t>& tag, const std::string& suffix);

// " [96K]" / " [88.2K]" for the -a switch
std::string album_rate_suffix(uint32_t rate);

struct TagPicture {
    uint32_t type = 3;               // ID3 APIC picture type == FLAC PICTURE type
    std::string mime, description;   // description in UTF-8
    std::vector<uint8_t> data;
};

// Text frames as (VORBIS_FIELD, utf-8 value) pairs, and the attached pictures.
void tag_to_vorbis(const std::vector<uint8_t>& tag, std::vector<std::pair<std::string, std::string>>& fields,
                   std::vector<TagPicture>& pictures);

// Copies the image files (jpg, jpeg, png, gif, bmp, tif, tiff, webp) of `from_dir` into `to_dir`;
// existing files of the same size are left alone.  Returns the number of files copied.
int copy_artwork(const std::string& from_dir, const std::string& to_dir);

}  // namespace d2dhost
