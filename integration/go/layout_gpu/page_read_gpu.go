//go:build pqgpu

// GPU-backed read side of package layout: drop this file into github.com/xitongsys/parquet-go/layout.
//
// It defines the four read functions reader/columnbuffer.go calls - ReadPage (layout/page.go:721),
// ReadPageRawData (:395), (*Page).GetRLDLFromRawData (:433), (*Page).GetValueFromRawData (:562) - with their
// signatures, results and error values unchanged, so reader/columnbuffer.go and reader/columnreader.go compile and
// run as they are.  The reference's own definitions of the four stay in the package under private names (readPageCPU,
// readPageRawDataCPU, getRLDLFromRawDataCPU, getValueFromRawDataCPU: a file page_read_cpu.go that is compiled under both
// tags) and this file hands every chunk it does not take to them: compressed chunks (the reference's writer
// compresses with SNAPPY unless told otherwise, writer/writer.go:63-65) and chunks holding a page the library answers
// with PQD_ERR_UNSUPPORTED_*, PQD_ERR_BIT_WIDTH, PQD_ERR_NO_DICTIONARY or PQD_ERR_REFUSED.  A build with the tag reads
// every file the reference reads.
// (*Page).Decode (:139) stays where it is: the pages handed out here already hold the looked-up values and say so
// in their header (Encoding PLAIN), which makes Decode return at its first test of the encoding.
//
// How it works: the reference reads and decodes one page per call.  Here the first call on a column chunk reads the
// whole chunk (ColumnMetaData.TotalCompressedSize bytes) from the thrift reader, has the GPU decode every page of it in
// one call of the C ABI and keeps the result; that call and the following ones each hand out one page of it, with the
// (numValues, numRows) the reference returns for that page.  The state is keyed by the thrift reader:
// ColumnBufferType opens a new one for every row group (reader/columnbuffer.go:95-99).
//
// Source only: written without a Go toolchain at hand and never compiled; parquet-go_amd/layout_gpu.py makes the same
// calls in the same order (the hand-over to the CPU functions included) and is what the tests drive.
//
// Lifetimes: a chunk's handle goes back to a bounded free list when the chunk's last page has been handed out, at
// io.EOF, or when the chunk goes over to the CPU functions - not in a finalizer.  State is keyed by the address of the
// thrift reader (and of pages handed out raw), because neither type can carry a field of ours without touching the
// reference's files; the entries are small Go structs once the handle is back, and finalizers remove them.  A host that
// wants nothing left behind between collections calls ReleaseReader(thriftReader) when it drops a reader.
package layout

import (
	"bytes"
	"fmt"
	"io"
	"os"
	"runtime"
	"strconv"
	"sync"
	"unsafe"

	"github.com/apache/thrift/lib/go/thrift"
	"github.com/xitongsys/parquet-go/common"
	"github.com/xitongsys/parquet-go/parquet"
	"github.com/xitongsys/parquet-go/pqdecode"
	"github.com/xitongsys/parquet-go/schema"
)

// decoded is the result of one C-ABI call over the pages [first, last) of a chunk.
type decoded struct {
	first, last int
	lead        int // 1 when the chunk's dictionary page was put in front of the descriptors
	res         *pqdecode.Result
	slot0       []int // first slot of every page of the call, and the total
}

func newDecoded(first, last, lead int, res *pqdecode.Result) *decoded {
	d := &decoded{first: first, last: last, lead: lead, res: res, slot0: make([]int, last-first+1)}
	for k := first; k < last; k++ {
		d.slot0[k-first+1] = d.slot0[k-first] + int(res.Counts[lead+k-first].Slots)
	}
	return d
}

func (d *decoded) covers(k int) bool { return d != nil && d.first <= k && k < d.last }

type pageError struct {
	page   int
	err    error
	status int
}

// chunkPager stands for the thrift reader positioned inside one column chunk.
type chunkPager struct {
	h        *pqdecode.Handle // nil once the chunk is done with the GPU (handed back deterministically, see release)
	chunk    []byte           // page-locked input buffer of the handle while h != nil; a Go copy (or nil) afterwards
	descs    []pqdecode.PageDesc
	col      pqdecode.Column
	cursor   int
	dictIdx  int   // dictionary page handed out last, -1: none
	dictPage *Page // that page as the caller holds it (ColumnBufferType.DictPage)
	full     *decoded
	levels   *decoded
	errFull  *pageError
	errLevel *pageError
	rawOut   int // pages handed out by ReadPageRawData whose values have not been asked for

	// cpu != nil: from here on the chunk's pages come from the reference's own functions reading this transport - the
	// caller's reader for a chunk the GPU never saw, a reader over the chunk's bytes in memory for one handed over part way
	cpu *thrift.TBufferedTransport

	path    []string
	repType parquet.FieldRepetitionType
	schema  *parquet.SchemaElement
	codec   parquet.CompressionCodec
	colMeta *parquet.ColumnMetaData
}

var (
	pagers      sync.Map // uintptr(thrift reader) -> *chunkPager
	rawPages    sync.Map // uintptr(*Page handed out by ReadPageRawData) -> rawRef
	freeHandles = make(chan *pqdecode.Handle, maxFreeHandles)
	gpuDevice   = func() int { d, _ := strconv.Atoi(os.Getenv("PQD_DEVICE")); return d }()
)

// Handles kept for reuse (each holds a HIP stream, grow-only device buffers and two page-locked staging buffers, none
// of which the Go collector sees): at most this many wait here, the rest are closed when their chunk is done.
const maxFreeHandles = 16

type rawRef struct {
	pg    *chunkPager
	index int
}

func takeHandle() (*pqdecode.Handle, error) {
	select {
	case h := <-freeHandles:
		return h, nil
	default:
		return pqdecode.NewHandle(gpuDevice) // (carries a finalizer that closes it, as a backstop)
	}
}

func putHandle(h *pqdecode.Handle) {
	select {
	case freeHandles <- h:
	default:
		h.Close()
	}
}

// delegable: codes on which the reference behaves in a way the typed GPU result cannot express (or simply fails in its
// own words); the chunk is handed to the reference's own functions from the failing page on, so the caller sees exactly
// what the reference does.
func delegable(status int) bool {
	switch status {
	case pqdecode.ErrUnsupportedEncoding, pqdecode.ErrUnsupportedType, pqdecode.ErrBitWidth, pqdecode.ErrNoDictionary, pqdecode.ErrRefused:
		return true
	}
	return false
}

// pagerFor returns the hand-out state of the chunk the thrift reader is positioned in, reading the chunk on the
// first call.  io.EOF when not a byte is left, as ReadPageHeader reports it (reader/columnbuffer.go:108-129 gives
// that error a meaning).
func pagerFor(thriftReader *thrift.TBufferedTransport, schemaHandler *schema.SchemaHandler, colMetaData *parquet.ColumnMetaData) (*chunkPager, error) {
	key := uintptr(unsafe.Pointer(thriftReader))
	if p, ok := pagers.Load(key); ok {
		return p.(*chunkPager), nil
	}
	// the map must not keep the reader alive: it is keyed by address, and the entry goes when the reader does
	register := func(pg *chunkPager) {
		pagers.Store(key, pg)
		runtime.SetFinalizer(thriftReader, func(t *thrift.TBufferedTransport) {
			pagers.Delete(uintptr(unsafe.Pointer(t)))
			pg.release(false)
		})
	}
	if colMetaData.GetCodec() != parquet.CompressionCodec_UNCOMPRESSED {
		// compressed pages are decompressed and decoded by the reference's own code (layout/page.go:753-758, :780-783)
		pg := &chunkPager{cpu: thriftReader, dictIdx: -1}
		register(pg)
		return pg, nil
	}
	h, err := takeHandle()
	if err != nil {
		return nil, err
	}
	buf, err := h.InputBuffer(int(colMetaData.GetTotalCompressedSize()))
	if err != nil {
		putHandle(h)
		return nil, err
	}
	n, err := io.ReadFull(thriftReader, buf)
	if n == 0 {
		putHandle(h)
		if err == nil {
			err = io.EOF
		}
		return nil, err
	}
	buf = buf[:n] // a chunk cut short by the end of the file: the page that runs past it fails, not the chunk
	descs, err := pqdecode.ScanPages(buf, colMetaData.GetNumValues())
	if err != nil && len(descs) == 0 {
		putHandle(h)
		return nil, err
	}
	path := append([]string{schemaHandler.GetRootInName()}, colMetaData.GetPathInSchema()...)
	name := common.PathToStr(path)
	maxDef, _ := schemaHandler.MaxDefinitionLevel(path)
	maxRep, _ := schemaHandler.MaxRepetitionLevel(path)
	el := schemaHandler.SchemaElements[schemaHandler.MapIndex[name]]
	pg := &chunkPager{h: h, chunk: buf, descs: descs, dictIdx: -1, path: path, repType: el.GetRepetitionType(), schema: el,
		codec: colMetaData.GetCodec(), colMeta: colMetaData,
		col:   pqdecode.Column{Type: colMetaData.GetType(), TypeLength: el.GetTypeLength(), MaxDef: maxDef, MaxRep: maxRep}}
	register(pg)
	return pg, nil
}

// release hands the chunk's handle back: when the chunk's last page has been handed out, when it goes over to the CPU
// functions, and (backstop) when the reader is collected.  keepBytes: pages handed out raw may still ask for their
// values (GetValueFromRawData), so the chunk's bytes are copied out of the handle's page-locked buffer first; such a
// call then borrows a handle for its own duration.
func (pg *chunkPager) release(keepBytes bool) {
	if pg.h == nil {
		return
	}
	if keepBytes {
		pg.chunk = append([]byte(nil), pg.chunk...)
	} else {
		pg.chunk = nil
	}
	putHandle(pg.h)
	pg.h = nil
}

// headerOffset: where the header of page k starts in the chunk (the end of the page before it).
func (pg *chunkPager) headerOffset(k int) int64 {
	if k == 0 {
		return 0
	}
	return pg.descs[k-1].Offset() + int64(pg.descs[k-1].Length())
}

func memReader(b []byte) *thrift.TBufferedTransport {
	return thrift.NewTBufferedTransport(thrift.NewStreamTransportR(bytes.NewReader(b)), 4096)
}

// toCPU hands the chunk over to the reference's functions from page k on: they read a copy of the chunk's bytes from
// that page's header.  A dictionary page handed out earlier went out with an empty table (its entries were used on the
// device); the pages the CPU decodes look their values up in it (Page.Decode, layout/page.go:139-164), so it is decoded
// now and filled in.
func (pg *chunkPager) toCPU(k int, schemaHandler *schema.SchemaHandler, colMetaData *parquet.ColumnMetaData) error {
	if pg.dictIdx >= 0 && pg.dictPage != nil {
		d, _, _, err := readPageCPU(memReader(pg.chunk[pg.headerOffset(pg.dictIdx):]), schemaHandler, colMetaData)
		if err != nil {
			return err
		}
		pg.dictPage.DataTable = d.DataTable
	}
	rest := append([]byte(nil), pg.chunk[pg.headerOffset(k):]...)
	pg.cpu = memReader(rest)
	pg.release(pg.rawOut > 0)
	return nil
}

// rawToCPU turns a page handed out by ReadPageRawData into the page the reference's ReadPageRawData would have handed
// out (header, compressed bytes and all), read from the chunk's bytes; the reference's two methods take it from there.
// The dictionary page the caller holds is filled in as in toCPU.
func (pg *chunkPager) rawToCPU(p *Page, k int, schemaHandler *schema.SchemaHandler) error {
	if pg.chunk == nil {
		return fmt.Errorf("pqdecode: the chunk's bytes have been released")
	}
	if pg.dictIdx >= 0 && pg.dictPage != nil && pg.dictPage.DataTable != nil && len(pg.dictPage.DataTable.Values) == 0 {
		d, _, _, err := readPageCPU(memReader(pg.chunk[pg.headerOffset(pg.dictIdx):]), schemaHandler, pg.colMeta)
		if err != nil {
			return err
		}
		pg.dictPage.DataTable = d.DataTable
	}
	q, err := readPageRawDataCPU(memReader(pg.chunk[pg.headerOffset(k):]), schemaHandler, pg.colMeta)
	if err != nil {
		return err
	}
	if _, ok := rawPages.LoadAndDelete(uintptr(unsafe.Pointer(p))); ok {
		pg.rawOut--
	}
	*p = *q
	return nil
}

// call decodes the pages [first, last), behind the dictionary page in effect when there is one.
func (pg *chunkPager) call(first, last int, levelsOnly bool) (*pqdecode.Result, int, error) {
	lead := 0
	descs := pg.descs[first:last]
	if pg.dictIdx >= 0 && !levelsOnly && pg.descs[first].PageType() != parquet.PageType_DICTIONARY_PAGE {
		lead = 1
		descs = append([]pqdecode.PageDesc{pg.descs[pg.dictIdx]}, descs...)
	}
	h := pg.h
	if h == nil { // the chunk's handle has gone back already (a raw page asking for its values late): borrow one
		if pg.chunk == nil {
			return nil, 0, fmt.Errorf("pqdecode: the chunk's bytes have been released")
		}
		var err error
		if h, err = takeHandle(); err != nil {
			return nil, 0, err
		}
		defer putHandle(h)
	}
	res, err := h.Decode(pg.chunk, descs, pg.col, levelsOnly)
	return res, lead, err
}

// decodeFrom decodes the pages from k to the end of the chunk; pages in front of a failing one stay available.
func (pg *chunkPager) decodeFrom(k int, levelsOnly bool) error {
	last := len(pg.descs)
	res, lead, err := pg.call(k, last, levelsOnly)
	if err != nil {
		return err
	}
	var perr *pageError
	if res.Status != 0 {
		bad := k + res.BadPage - lead
		if lead == 1 && res.BadPage == 0 {
			bad = k // the dictionary page handed out earlier failed: the next page meets it
		}
		perr = &pageError{bad, pqdecode.Err(res.Status, bad), res.Status}
		last = bad
		if last > k {
			if res, lead, err = pg.call(k, last, levelsOnly); err != nil {
				return err
			}
			if res.Status != 0 {
				return pqdecode.Err(res.Status, k+res.BadPage-lead)
			}
		}
	}
	var d *decoded
	if last > k {
		d = newDecoded(k, last, lead, res)
	}
	if levelsOnly {
		pg.levels, pg.errLevel = d, perr
	} else {
		pg.full, pg.errFull = d, perr
		if perr == nil {
			pg.levels = d // a full decode without failures serves the level calls too
		}
	}
	return nil
}

// table builds the layout.Table of page k out of a call's result (layout/page.go:905-931).
func (pg *chunkPager) table(d *decoded, k int, withValues bool) (*Table, int64, int64) {
	a, b := d.slot0[k-d.first], d.slot0[k-d.first+1]
	t := new(Table)
	t.Path = pg.path
	t.RepetitionType = pg.repType
	t.MaxDefinitionLevel, t.MaxRepetitionLevel = pg.col.MaxDef, pg.col.MaxRep
	t.DefinitionLevels = d.res.Def[a:b:b]
	t.RepetitionLevels = d.res.Rep[a:b:b]
	if withValues && d.res.Values != nil {
		t.Values = d.res.Values[a:b:b]
	} else {
		t.Values = make([]interface{}, b-a)
	}
	return t, int64(b - a), d.res.Counts[d.lead+k-d.first].Rows
}

// header is the parquet.PageHeader of page k as far as the reader's callers look at it.  Encoding is PLAIN for every
// data page: its table holds values, never dictionary indices.
func (pg *chunkPager) header(k int) *parquet.PageHeader {
	d := &pg.descs[k]
	h := parquet.NewPageHeader()
	h.Type = d.PageType()
	h.CompressedPageSize, h.UncompressedPageSize = d.Length(), d.Length()
	switch d.PageType() {
	case parquet.PageType_DICTIONARY_PAGE:
		h.DictionaryPageHeader = parquet.NewDictionaryPageHeader()
		h.DictionaryPageHeader.NumValues = d.NumValues()
		h.DictionaryPageHeader.Encoding = parquet.Encoding_PLAIN
	default:
		h.DataPageHeader = parquet.NewDataPageHeader()
		h.DataPageHeader.NumValues = d.NumValues()
		h.DataPageHeader.Encoding = parquet.Encoding_PLAIN
		h.DataPageHeader.DefinitionLevelEncoding = parquet.Encoding_RLE
		h.DataPageHeader.RepetitionLevelEncoding = parquet.Encoding_RLE
	}
	return h
}

func (pg *chunkPager) newPage(k int) *Page {
	var page *Page
	if pg.descs[k].PageType() == parquet.PageType_DICTIONARY_PAGE {
		page = NewDictPage()
	} else {
		page = NewDataPage()
	}
	page.Header = pg.header(k)
	page.CompressType = pg.codec
	page.Path = pg.path
	page.Schema = pg.schema
	return page
}

func failed(e *pageError, k int) bool { return e != nil && k >= e.page }

// ReleaseReader drops what this file keeps for a thrift reader (its chunk's handle, bytes and results) at once instead of
// when the reader is collected.  Optional: reader/columnbuffer.go does not call it and need not.
func ReleaseReader(thriftReader *thrift.TBufferedTransport) {
	key := uintptr(unsafe.Pointer(thriftReader))
	if p, ok := pagers.LoadAndDelete(key); ok {
		p.(*chunkPager).release(false)
	}
}

// ReadPage hands out the next page of the chunk: (page, numValues, numRows, err) as layout/page.go:721-937 returns
// them.  A dictionary page comes back with an empty table (its entries are used on the device) and zeros.
func ReadPage(thriftReader *thrift.TBufferedTransport, schemaHandler *schema.SchemaHandler, colMetaData *parquet.ColumnMetaData) (*Page, int64, int64, error) {
	pg, err := pagerFor(thriftReader, schemaHandler, colMetaData)
	if err != nil {
		return nil, 0, 0, err
	}
	if pg.cpu != nil {
		return readPageCPU(pg.cpu, schemaHandler, colMetaData)
	}
	k := pg.cursor
	if k >= len(pg.descs) {
		pg.release(pg.rawOut > 0)
		return nil, 0, 0, io.EOF
	}
	switch pg.descs[k].PageType() {
	case parquet.PageType_DATA_PAGE, parquet.PageType_DATA_PAGE_V2, parquet.PageType_DICTIONARY_PAGE:
	case parquet.PageType_INDEX_PAGE:
		return nil, 0, 0, fmt.Errorf("Unsupported page type: INDEX_PAGE")
	default:
		return nil, 0, 0, fmt.Errorf("Error page type %v", pg.descs[k].PageType())
	}
	if !pg.full.covers(k) && !failed(pg.errFull, k) {
		if err = pg.decodeFrom(k, false); err != nil {
			return nil, 0, 0, err
		}
	}
	if failed(pg.errFull, k) {
		if delegable(pg.errFull.status) {
			// the reference decodes (or refuses) this page in its own way: from here on the chunk is its own
			if err = pg.toCPU(k, schemaHandler, colMetaData); err != nil {
				return nil, 0, 0, err
			}
			return readPageCPU(pg.cpu, schemaHandler, colMetaData)
		}
		return nil, 0, 0, pg.errFull.err
	}
	pg.cursor++
	page := pg.newPage(k)
	if pg.descs[k].PageType() == parquet.PageType_DICTIONARY_PAGE {
		pg.dictIdx, pg.dictPage = k, page
		page.DataTable = &Table{Path: pg.path}
		return page, 0, 0, nil
	}
	var numValues, numRows int64
	page.DataTable, numValues, numRows = pg.table(pg.full, k, true)
	if pg.cursor == len(pg.descs) {
		// the chunk's last page: its table lives in Go memory, so the handle can serve another chunk right away
		pg.release(pg.rawOut > 0)
	}
	return page, numValues, numRows, nil
}

// ReadPageRawData hands out the next page undecoded (layout/page.go:395-430); GetRLDLFromRawData and
// GetValueFromRawData find their way back to the chunk through the page.
func ReadPageRawData(thriftReader *thrift.TBufferedTransport, schemaHandler *schema.SchemaHandler, colMetaData *parquet.ColumnMetaData) (*Page, error) {
	pg, err := pagerFor(thriftReader, schemaHandler, colMetaData)
	if err != nil {
		return nil, err
	}
	if pg.cpu != nil {
		return readPageRawDataCPU(pg.cpu, schemaHandler, colMetaData) // (its pages are not in rawPages: the two methods below pass them on)
	}
	k := pg.cursor
	if k >= len(pg.descs) {
		pg.release(pg.rawOut > 0)
		return nil, io.EOF
	}
	switch pg.descs[k].PageType() {
	case parquet.PageType_DATA_PAGE, parquet.PageType_DATA_PAGE_V2, parquet.PageType_DICTIONARY_PAGE:
	default:
		return nil, fmt.Errorf("Unsupported page type")
	}
	pg.cursor++
	page := pg.newPage(k)
	key := uintptr(unsafe.Pointer(page))
	rawPages.Store(key, rawRef{pg, k})
	pg.rawOut++
	// (the entry is removed when the page's values have been fetched; the finalizer covers pages that are skipped for good)
	runtime.SetFinalizer(page, func(p *Page) { rawPages.Delete(uintptr(unsafe.Pointer(p))) })
	return page, nil
}

// GetRLDLFromRawData: levels, (numValues, numRows) of the page; values left undecoded (layout/page.go:433-559).
func (p *Page) GetRLDLFromRawData(schemaHandler *schema.SchemaHandler) (int64, int64, error) {
	ref, ok := rawPages.Load(uintptr(unsafe.Pointer(p)))
	if !ok {
		return p.getRLDLFromRawDataCPU(schemaHandler) // a page of a chunk the CPU functions read
	}
	pg, k := ref.(rawRef).pg, ref.(rawRef).index
	if pg.descs[k].PageType() == parquet.PageType_DICTIONARY_PAGE {
		pg.dictIdx, pg.dictPage = k, p // ReadPageForSkip keeps it as DictPage (reader/columnbuffer.go:170-174)
		p.DataTable = &Table{Path: pg.path}
		return 0, 0, nil
	}
	if !pg.levels.covers(k) && !failed(pg.errLevel, k) {
		if err := pg.decodeFrom(k, true); err != nil {
			return 0, 0, err
		}
	}
	if failed(pg.errLevel, k) {
		if delegable(pg.errLevel.status) {
			if err := pg.rawToCPU(p, k, schemaHandler); err != nil {
				return 0, 0, err
			}
			return p.getRLDLFromRawDataCPU(schemaHandler)
		}
		return 0, 0, pg.errLevel.err
	}
	var numValues, numRows int64
	p.DataTable, numValues, numRows = pg.table(pg.levels, k, false)
	return numValues, numRows, nil
}

// GetValueFromRawData: the values of a page whose levels are known (layout/page.go:562-622): the page alone, behind
// its dictionary page, in one call - ColumnBufferType.SkipRows asks for the last page it read only.
func (p *Page) GetValueFromRawData(schemaHandler *schema.SchemaHandler) error {
	ref, ok := rawPages.Load(uintptr(unsafe.Pointer(p)))
	if !ok {
		return p.getValueFromRawDataCPU(schemaHandler)
	}
	pg, k := ref.(rawRef).pg, ref.(rawRef).index
	rawPages.Delete(uintptr(unsafe.Pointer(p)))
	pg.rawOut--
	if pg.descs[k].PageType() == parquet.PageType_DICTIONARY_PAGE {
		return nil // its entries are decoded with the data pages that use them
	}
	d := pg.full
	if !d.covers(k) {
		res, lead, err := pg.call(k, k+1, false)
		if err != nil {
			return err
		}
		if res.Status != 0 {
			if delegable(res.Status) {
				// the page as the reference reads it, its levels decoded once more by the reference's code, then its values
				keep := p.DataTable
				if err := pg.rawToCPU(p, k, schemaHandler); err != nil {
					return err
				}
				if _, _, err := p.getRLDLFromRawDataCPU(schemaHandler); err != nil {
					return err
				}
				if err := p.getValueFromRawDataCPU(schemaHandler); err != nil {
					return err
				}
				if keep != nil { // the table the caller has merged its levels from stays the page's table
					keep.Values = p.DataTable.Values
					p.DataTable = keep
				}
				return nil
			}
			return pqdecode.Err(res.Status, k)
		}
		d = newDecoded(k, k+1, lead, res)
	}
	t, _, _ := pg.table(d, k, true)
	// keep the table GetRLDLFromRawData made (the caller has merged its levels already): fill in the values
	if p.DataTable == nil {
		p.DataTable = t
	} else {
		p.DataTable.Values = t.Values
	}
	p.RawData = []byte{}
	return nil
}
