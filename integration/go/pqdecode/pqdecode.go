// Package pqdecode binds libpqdecode.so (include/pqdecode.h) for xitongsys/parquet-go.
//
// Source only: no Go toolchain exists in the image this library is built and tested in, so this file has not been
// compiled there.  The Python ctypes mirror (parquet-go_amd/binding.py, parquet-go_amd/columnbuffer.py) makes the
// same calls in the same order and is what the parity tests drive.
//
// Build:  CGO_CFLAGS=-I<repo>/include  CGO_LDFLAGS="-L<repo>/parquet-go_amd/lib -lpqdecode"  go build ./...
//
// What it replaces: the per-page loop of reader.ColumnBufferType.ReadPage (reader/columnbuffer.go:105-156:
// layout.ReadPage + Page.Decode + Table.Merge for every page) by one call per column chunk.
package pqdecode

/*
#include <stdlib.h>
#include "pqdecode.h"
*/
import "C"

import (
	"fmt"
	"io"
	"math"
	"unsafe"

	"github.com/xitongsys/parquet-go/layout"
	"github.com/xitongsys/parquet-go/parquet"
)

// Handle is one HIP stream plus device workspaces; one per reader.ColumnBufferType, used by one goroutine at a
// time (reader/reader.go:306-327 drives every column buffer from its own goroutine).
type Handle struct{ h *C.pqd_handle }

func NewHandle(device int) (*Handle, error) {
	var h *C.pqd_handle
	if rc := C.pqd_handle_create(C.int(device), nil, &h); rc != 0 {
		return nil, libErr(rc, -1)
	}
	return &Handle{h}, nil
}

func (h *Handle) Close() { C.pqd_handle_destroy(h.h); h.h = nil }

// libErr maps the library's codes onto the errors the reference's callers test for
// (io.EOF has meaning to ColumnBufferType.ReadPage, reader/columnbuffer.go:110).
func libErr(code C.int, page C.int) error {
	switch code {
	case C.PQD_OK:
		return nil
	case C.PQD_ERR_EOF:
		return io.EOF
	case C.PQD_ERR_UNEXPECTED_EOF:
		return io.ErrUnexpectedEOF
	case C.PQD_ERR_UNSUPPORTED_ENCODING:
		return fmt.Errorf("Unknown Encoding method (page %d)", int(page)) // layout/page.go:716
	case C.PQD_ERR_UNSUPPORTED_PAGE_TYPE:
		return fmt.Errorf("Unsupported page type (page %d)", int(page)) // layout/page.go:813
	default:
		return fmt.Errorf("pqdecode: %s (page %d): %s", C.GoString(C.pqd_strerror(code)), int(page), C.GoString(C.pqd_last_error()))
	}
}

// Column is what layout.ReadPage takes from the schema handler and the column metadata for one leaf
// (layout/page.go:797-805, :821-822, :897-900).
type Column struct {
	Path       []string
	Type       parquet.Type
	TypeLength int32
	MaxDef     int32
	MaxRep     int32
}

// DecodeChunk decodes all pages of one column chunk: chunk holds the chunk's bytes as they are in the file, from
// the first page header on (reader/columnbuffer.go:89-99 gives the offset; MetaData.TotalCompressedSize the
// length); the codec must be UNCOMPRESSED.  It returns the table ColumnBufferType.ReadPage would have merged
// together page by page, and the number of row starts in it.
//
// chunk is copied into pinned staging inside pqd_decode_chunk, so no Go pointer is retained after the call
// returns (the cgo pointer rule).
func (h *Handle) DecodeChunk(chunk []byte, numValues int64, col Column) (*layout.Table, int64, error) {
	t := layout.NewEmptyTable()
	t.Path, t.MaxDefinitionLevel, t.MaxRepetitionLevel = col.Path, col.MaxDef, col.MaxRep
	if len(chunk) == 0 {
		return t, 0, nil
	}
	p := (*C.uint8_t)(unsafe.Pointer(&chunk[0]))
	descs := make([]C.pqd_page_desc, 64)
	var n C.int32_t
	for {
		rc := C.pqd_scan_pages(p, C.size_t(len(chunk)), C.int64_t(numValues), &descs[0], C.int32_t(len(descs)), &n)
		if rc == C.PQD_ERR_NOMEM { // more pages than descriptors: grow and walk again
			descs = make([]C.pqd_page_desc, 4*len(descs))
			continue
		}
		if rc != 0 {
			return nil, 0, libErr(rc, C.int(n))
		}
		break
	}
	cd := C.pqd_column_desc{physical_type: C.int32_t(col.Type), type_length: C.int32_t(col.TypeLength),
		max_def: C.int32_t(col.MaxDef), max_rep: C.int32_t(col.MaxRep)}
	if rc := C.pqd_decode_chunk(h.h, p, C.size_t(len(chunk)), &descs[0], n, &cd); rc != 0 {
		return nil, 0, libErr(rc, -1)
	}
	var res C.pqd_chunk_result
	if rc := C.pqd_wait(h.h, &res); rc != 0 {
		return nil, 0, libErr(rc, -1)
	}
	if res.status != 0 {
		return nil, 0, libErr(C.int(res.status), C.int(res.bad_page))
	}
	slots := int(res.num_slots)
	t.DefinitionLevels = make([]int32, slots)
	t.RepetitionLevels = make([]int32, slots)
	t.Values = make([]interface{}, slots)
	if slots == 0 {
		return t, 0, nil
	}
	// The result is read in place from the handle's page-locked memory (valid until the handle's next decode call):
	// the level arrays are copied into the Table, the values are boxed straight from it.
	var hr C.pqd_device_result
	if rc := C.pqd_result_host(h.h, &hr); rc != 0 {
		return nil, 0, libErr(rc, -1)
	}
	if hr.def_levels != nil {
		copy(t.DefinitionLevels, unsafe.Slice((*int32)(unsafe.Pointer(hr.def_levels)), slots))
	}
	if hr.rep_levels != nil {
		copy(t.RepetitionLevels, unsafe.Slice((*int32)(unsafe.Pointer(hr.rep_levels)), slots))
	}
	if col.Type == parquet.Type_BYTE_ARRAY {
		offs := unsafe.Slice((*int64)(unsafe.Pointer(hr.offsets)), slots+1)
		buf := unsafe.Slice((*byte)(unsafe.Pointer(hr.bytes)), int(res.bytes_len))
		for i := 0; i < slots; i++ {
			if t.DefinitionLevels[i] == col.MaxDef {
				t.Values[i] = string(buf[offs[i]:offs[i+1]]) // copies; the reference holds BYTE_ARRAY values as string
			}
		}
		return t, int64(res.num_rows), nil
	}
	esz := int(res.elem_size)
	raw := unsafe.Slice((*byte)(hr.values), slots*esz)
	box(t, raw, esz, col)
	return t, int64(res.num_rows), nil
}

// InputBuffer returns n bytes of the handle's page-locked input buffer.  A caller that reads the chunk from the file
// straight into it (io.ReadFull(pFile, buf)) and passes the same slice to DecodeChunk saves one copy of the chunk;
// the memory belongs to the library, so no Go pointer is involved.  Fill it only when the handle's previous
// DecodeChunk has returned.
func (h *Handle) InputBuffer(n int) ([]byte, error) {
	var p *C.uint8_t
	if rc := C.pqd_input_buffer(h.h, C.size_t(n), &p); rc != 0 {
		return nil, libErr(rc, -1)
	}
	return unsafe.Slice((*byte)(unsafe.Pointer(p)), n), nil
}

// box fills Table.Values with the boxed Go values the reference's decoders produce (encoding/encodingread.go:13-125:
// bool, int32, int64, string of 12 bytes for INT96, float32, float64, string for FIXED_LEN_BYTE_ARRAY), nil where the
// definition level is below the maximum (layout/page.go:905-923).
func box(t *layout.Table, raw []byte, esz int, col Column) {
	le32 := func(b []byte) uint32 { return uint32(b[0]) | uint32(b[1])<<8 | uint32(b[2])<<16 | uint32(b[3])<<24 }
	le64 := func(b []byte) uint64 { return uint64(le32(b)) | uint64(le32(b[4:]))<<32 }
	for i := range t.Values {
		if t.DefinitionLevels[i] != col.MaxDef {
			continue
		}
		b := raw[i*esz : (i+1)*esz]
		switch col.Type {
		case parquet.Type_BOOLEAN:
			t.Values[i] = b[0] != 0
		case parquet.Type_INT32:
			t.Values[i] = int32(le32(b))
		case parquet.Type_INT64:
			t.Values[i] = int64(le64(b))
		case parquet.Type_FLOAT:
			t.Values[i] = math.Float32frombits(le32(b))
		case parquet.Type_DOUBLE:
			t.Values[i] = math.Float64frombits(le64(b))
		default: // INT96, FIXED_LEN_BYTE_ARRAY
			t.Values[i] = string(b)
		}
	}
}

// SkipLevels is the decode ColumnBufferType.SkipRows needs (reader/columnbuffer.go:194-230): levels and the row
// count of the chunk, values left undecoded.
func (h *Handle) SkipLevels(chunk []byte, numValues int64, col Column) (dls, rls []int32, numRows int64, err error) {
	if len(chunk) == 0 {
		return nil, nil, 0, nil
	}
	p := (*C.uint8_t)(unsafe.Pointer(&chunk[0]))
	descs := make([]C.pqd_page_desc, 1+len(chunk)/64)
	var n C.int32_t
	if rc := C.pqd_scan_pages(p, C.size_t(len(chunk)), C.int64_t(numValues), &descs[0], C.int32_t(len(descs)), &n); rc != 0 {
		return nil, nil, 0, libErr(rc, C.int(n))
	}
	cd := C.pqd_column_desc{physical_type: C.int32_t(col.Type), type_length: C.int32_t(col.TypeLength),
		max_def: C.int32_t(col.MaxDef), max_rep: C.int32_t(col.MaxRep)}
	if rc := C.pqd_decode_levels(h.h, p, C.size_t(len(chunk)), &descs[0], n, &cd); rc != 0 {
		return nil, nil, 0, libErr(rc, -1)
	}
	var res C.pqd_chunk_result
	if rc := C.pqd_wait(h.h, &res); rc != 0 {
		return nil, nil, 0, libErr(rc, -1)
	}
	if res.status != 0 {
		return nil, nil, 0, libErr(C.int(res.status), C.int(res.bad_page))
	}
	slots := int(res.num_slots)
	dls, rls = make([]int32, slots), make([]int32, slots)
	if slots > 0 {
		if rc := C.pqd_copy_out(h.h, nil, (*C.int32_t)(unsafe.Pointer(&dls[0])), (*C.int32_t)(unsafe.Pointer(&rls[0])), nil, nil); rc != 0 {
			return nil, nil, 0, libErr(rc, -1)
		}
	}
	return dls, rls, int64(res.num_rows), nil
}
